"""Per-network default options (reference dlib/utils/utils_init_default_args.py:13-50)."""
from copy import deepcopy

from dlib.utils import constants
from dlib.utils.shared import safe_str_var


def init_net_g(netG: dict, args: dict) -> dict:
    out = deepcopy(netG)
    nt = safe_str_var(netG['net_type'])
    if netG['net_type'] == constants.SWINIR:
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_img_size': args['h_size'] // args['scale'], f'{nt}_window_size': 8,
                    f'{nt}_img_range': 1.0, f'{nt}_depths': [6, 6, 6, 6, 6, 6], f'{nt}_embed_dim': 180,
                    f'{nt}_num_heads': [6, 6, 6, 6, 6, 6], f'{nt}_mlp_ratio': 2,
                    f'{nt}_upsampler': constants.US_PIXEL_SHUFFLE,
                    f'{nt}_resi_connection': constants.R_CONNECTION_1CONV})
    elif netG['net_type'] == constants.EDSR_LIIF:
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_n_feats': 64, f'{nt}_img_range': 1.0, f'{nt}_res_scale': 1.,
                    f'{nt}_n_resblocks': 16})
    elif netG['net_type'] == constants.DRRN:         # utils_init_default_args.py:231-235
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_num_residual_units': 25})
    elif netG['net_type'] == constants.VDSR:         # utils_init_default_args.py:241
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels']})
    elif netG['net_type'] == constants.SRCNN:        # utils_init_default_args.py (SRCNN: in_chans only)
        out.update({f'{nt}_in_chans': args['n_channels']})
    elif netG['net_type'] == constants.MSLAPSR:      # utils_init_default_args.py:118-125
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels']})
    elif netG['net_type'] == constants.MEMNET:       # utils_init_default_args.py:220-229
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_num_memory_blocks': 6, f'{nt}_num_residual_blocks': 6})
    elif netG['net_type'] == constants.DBPN:         # utils_init_default_args.py:153-163
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_base_filter': 64, f'{nt}_feat': 256, f'{nt}_num_stages': 3})
    elif netG['net_type'] == constants.SRFBN:        # utils_init_default_args.py:104-112
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_num_features': 64, f'{nt}_num_steps': 4, f'{nt}_num_groups': 6, f'{nt}_use_cl': True})
    elif netG['net_type'] == constants.PROSR:        # utils_init_default_args.py:127-151 (level_config hard-coded per scale)
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_residual_denseblock': True, f'{nt}_num_init_features': 160, f'{nt}_bn_size': 4,
                    f'{nt}_growth_rate': 40, f'{nt}_ps_woReLU': False, f'{nt}_level_compression': -1,
                    f'{nt}_res_factor': 0.2, f'{nt}_max_num_feature': 312, f'{nt}_block_compression': 0.4,
                    f'{nt}_level_config': {2: [[8] * 9], 4: [[8] * 9, [8] * 3], 8: [[8] * 9, [8] * 3, [8]]}})
    elif netG['net_type'] == constants.ENLCN:        # utils_init_default_args.py:74-82
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'], f'{nt}_n_resblock': 32,
                    f'{nt}_n_feats': 256, f'{nt}_res_scale': 0.1, f'{nt}_img_range': 1.0})
    elif netG['net_type'] == constants.NLSN:         # utils_init_default_args.py:88-98
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'], f'{nt}_n_resblocks': 32,
                    f'{nt}_n_feats': 256, f'{nt}_n_hashes': 4, f'{nt}_chunk_size': 144, f'{nt}_res_scale': 0.1,
                    f'{nt}_img_range': 1.0})
    elif netG['net_type'] == constants.DFCAN:        # utils_init_default_args.py:194-197
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels']})
    elif netG['net_type'] == constants.ACT:          # utils_init_default_args.py:52-66
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'], f'{nt}_n_feats': 64,
                    f'{nt}_img_range': 1.0, f'{nt}_n_resgroups': 4, f'{nt}_n_resblocks': 12, f'{nt}_reduction': 16,
                    f'{nt}_n_heads': 8, f'{nt}_n_layers': 8, f'{nt}_n_fusionblocks': 4, f'{nt}_dropout_rate': 0.0,
                    f'{nt}_token_size': 3, f'{nt}_expansion_ratio': 4})
    elif netG['net_type'] == constants.OMNISR:       # utils_init_default_args.py:203-214
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'], f'{nt}_num_feat': 64,
                    f'{nt}_res_num': 5, f'{nt}_bias': True, f'{nt}_window_size': 8, f'{nt}_block_num': 4, f'{nt}_pe': True,
                    f'{nt}_ffn_bias': True})
    elif netG['net_type'] == constants.GRL:          # utils_init_default_args.py:166-189 ("Big")
        out.update({f'{nt}_upscale': args['scale'], f'{nt}_in_chans': args['n_channels'],
                    f'{nt}_img_size': args['h_size'] // args['scale'], f'{nt}_window_size': 8, f'{nt}_embed_dim': 180,
                    f'{nt}_mlp_ratio': 2, f'{nt}_img_range': 1.0, f'{nt}_depths': [4, 4, 8, 8, 8, 4, 4],
                    f'{nt}_num_heads_window': [3, 3, 3, 3, 3, 3, 3], f'{nt}_num_heads_stripe': [3, 3, 3, 3, 3, 3, 3],
                    f'{nt}_upsampler': constants.US_PIXEL_SHUFFLE, f'{nt}_conv_type': '1conv', f'{nt}_out_proj_type': 'linear',
                    f'{nt}_anchor_window_down_factor': 2, f'{nt}_qkv_proj_type': 'linear', f'{nt}_anchor_proj_type': 'avgpool',
                    f'{nt}_local_connection': True})
    else:
        raise NotImplementedError(netG['net_type'])
    out[f'{nt}_init_type'] = constants.INIT_W_DEFAULT
    out[f'{nt}_init_bn_type'] = constants.INIT_BN_CONSTANT
    out[f'{nt}_init_gain'] = 1.
    return out
