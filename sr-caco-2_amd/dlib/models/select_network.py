"""Network registry: ``define_G(args) -> nn.Module`` keyed by
``args.netG['net_type']`` with per-net options read from
``args.netG[f'{safe_str_var(net_type)}_<opt>']`` -- the reference's contract
(dlib/models/select_network.py:19-50) for the two hot-path families."""
from dlib.utils import constants
from dlib.utils.shared import safe_str_var

__all__ = ['define_G']


def define_G(args):
    opt_net = args.netG
    net_type = opt_net['net_type']
    nt = safe_str_var(net_type)
    if net_type == constants.SWINIR:
        from dlib.models.network_swinir import SwinIR as net
        return net(upscale=opt_net[f'{nt}_upscale'], in_chans=opt_net[f'{nt}_in_chans'],
                   img_size=opt_net[f'{nt}_img_size'], window_size=opt_net[f'{nt}_window_size'],
                   img_range=opt_net[f'{nt}_img_range'], depths=opt_net[f'{nt}_depths'],
                   embed_dim=opt_net[f'{nt}_embed_dim'], num_heads=opt_net[f'{nt}_num_heads'],
                   mlp_ratio=opt_net[f'{nt}_mlp_ratio'], upsampler=opt_net[f'{nt}_upsampler'],
                   resi_connection=opt_net[f'{nt}_resi_connection'])
    if net_type == constants.EDSR_LIIF:
        from dlib.models.network_edsr_liif import EDSR_LIIF as net
        return net(in_chans=opt_net[f'{nt}_in_chans'], n_resblocks=opt_net[f'{nt}_n_resblocks'],
                   n_feats=opt_net[f'{nt}_n_feats'], scale=opt_net[f'{nt}_upscale'],
                   rgb_range=opt_net[f'{nt}_img_range'], res_scale=opt_net.get(f'{nt}_res_scale', 1.),
                   local_ensemble=True, feat_unfold=True, cell_decode=True)
    if net_type == constants.DRRN:                  # select_network.py:192-198
        from dlib.models.network_drrn import DRRN as net
        return net(in_chans=opt_net[f'{nt}_in_chans'], upscale=opt_net[f'{nt}_upscale'],
                   num_residual_units=opt_net[f'{nt}_num_residual_units'])
    if net_type == constants.VDSR:                  # select_network.py:200-205
        from dlib.models.network_vdsr import VDSR as net
        return net(in_chans=opt_net[f'{nt}_in_chans'], upscale=opt_net[f'{nt}_upscale'])
    if net_type == constants.SRCNN:                 # select_network.py:207-210
        from dlib.models.network_srcnn import SRCNN as net
        return net(in_chans=opt_net[f'{nt}_in_chans'])
    if net_type == constants.MSLAPSR:               # select_network.py:103-108
        from dlib.models.network_mslapsr import MSLapSRN as net
        return net(upscale=opt_net[f'{nt}_upscale'], in_chans=opt_net[f'{nt}_in_chans'])
    if net_type == constants.MEMNET:                # select_network.py:184-190
        from dlib.models.network_memnet import MemNet as net
        return net(in_chans=opt_net[f'{nt}_in_chans'], upscale=opt_net[f'{nt}_upscale'],
                   num_memory_blocks=opt_net[f'{nt}_num_memory_blocks'],
                   num_residual_blocks=opt_net[f'{nt}_num_residual_blocks'])
    if net_type == constants.DBPN:                  # select_network.py:139-147
        from dlib.models.network_dbpn import DBPN as net
        return net(upscale=opt_net[f'{nt}_upscale'], in_chans=opt_net[f'{nt}_in_chans'],
                   base_filter=opt_net[f'{nt}_base_filter'], feat=opt_net[f'{nt}_feat'],
                   num_stages=opt_net[f'{nt}_num_stages'])
    if net_type == constants.SRFBN:                 # select_network.py:130-137
        from dlib.models.network_srfbn import SRFBN as net
        return net(upscale=opt_net[f'{nt}_upscale'], in_chans=opt_net[f'{nt}_in_chans'],
                   num_features=opt_net[f'{nt}_num_features'], num_steps=opt_net[f'{nt}_num_steps'],
                   num_groups=opt_net[f'{nt}_num_groups'])
    if net_type == constants.ENLCN:                 # select_network.py:92-101
        from dlib.models.network_enlcn import ENLCN as net
        return net(upscale=opt_net[f'{nt}_upscale'], n_resblock=opt_net[f'{nt}_n_resblock'],
                   n_feats=opt_net[f'{nt}_n_feats'], res_scale=opt_net[f'{nt}_res_scale'],
                   img_range=opt_net[f'{nt}_img_range'], in_chans=opt_net[f'{nt}_in_chans'])
    if net_type == constants.NLSN:                  # select_network.py:149-160
        from dlib.models.network_nlsn import NLSN as net
        return net(upscale=opt_net[f'{nt}_upscale'], n_resblocks=opt_net[f'{nt}_n_resblocks'],
                   n_feats=opt_net[f'{nt}_n_feats'], n_hashes=opt_net[f'{nt}_n_hashes'],
                   chunk_size=opt_net[f'{nt}_chunk_size'], res_scale=opt_net[f'{nt}_res_scale'],
                   img_range=opt_net[f'{nt}_img_range'], in_chans=opt_net[f'{nt}_in_chans'])
    if net_type == constants.DFCAN:                 # select_network.py:162-167
        from dlib.models.network_dfcan import DFCAN as net
        return net(input_shape=opt_net[f'{nt}_in_chans'], upscale=opt_net[f'{nt}_upscale'])
    if net_type == constants.ACT:                   # select_network.py:52-68
        from dlib.models.network_act import ACT as net
        return net(**{k: opt_net[f'{nt}_{k}'] for k in ('upscale', 'in_chans', 'img_range', 'n_feats', 'n_resgroups',
                                                        'n_resblocks', 'reduction', 'n_heads', 'n_layers', 'dropout_rate',
                                                        'n_fusionblocks', 'token_size', 'expansion_ratio')})
    if net_type == constants.OMNISR:                # select_network.py:169-181
        from dlib.models.network_omni_sr import OmniSR as net
        return net(input_shape=opt_net[f'{nt}_in_chans'], upscale=opt_net[f'{nt}_upscale'],
                   num_feat=opt_net[f'{nt}_num_feat'], res_num=opt_net[f'{nt}_res_num'], bias=opt_net[f'{nt}_bias'],
                   window_size=opt_net[f'{nt}_window_size'], block_num=opt_net[f'{nt}_block_num'], pe=opt_net[f'{nt}_pe'],
                   ffn_bias=opt_net[f'{nt}_ffn_bias'])
    if net_type == constants.GRL:                   # select_network.py:70-90
        from dlib.models.network_grl import GRL as net
        return net(**{k: opt_net[f'{nt}_{k}'] for k in ('upscale', 'in_chans', 'img_size', 'window_size', 'img_range', 'depths',
                                                        'embed_dim', 'num_heads_window', 'num_heads_stripe', 'mlp_ratio',
                                                        'upsampler', 'qkv_proj_type', 'anchor_proj_type',
                                                        'anchor_window_down_factor', 'out_proj_type', 'conv_type',
                                                        'local_connection')})
    if net_type == constants.PROSR:                 # select_network.py:110-128
        from dlib.models.network_prosr import ProSR as net
        upscale = opt_net[f'{nt}_upscale']
        return net(upscale=upscale, in_chans=opt_net[f'{nt}_in_chans'],
                   residual_denseblock=opt_net[f'{nt}_residual_denseblock'],
                   num_init_features=opt_net[f'{nt}_num_init_features'], bn_size=opt_net[f'{nt}_bn_size'],
                   growth_rate=opt_net[f'{nt}_growth_rate'], ps_woReLU=opt_net[f'{nt}_ps_woReLU'],
                   level_config=opt_net[f'{nt}_level_config'][upscale],
                   level_compression=opt_net[f'{nt}_level_compression'], res_factor=opt_net[f'{nt}_res_factor'],
                   max_num_feature=opt_net[f'{nt}_max_num_feature'],
                   block_compression=opt_net[f'{nt}_block_compression'])
    raise NotImplementedError(
        f"net_type {net_type!r}: only {constants.MODELS} run on libsrhip (SURVEY.md section 8f lists "
        f"the remaining reference networks as 'next')")
