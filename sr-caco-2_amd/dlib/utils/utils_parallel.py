"""Eval-side collectives (reference dlib/utils/utils_parallel.py:13-64, used at
dlib/utils/utils_trainer.py:653-674): metric sums of the rank-sharded
validation / test pass gathered over the data-parallel group.

Same names and results as the reference.  Differences on purpose:
  * tensors stay where they are -- on the GPU the collective is RCCL over xGMI
    (``torch.distributed`` backend "nccl"), on CPU tensors it is gloo (tests);
    nothing here picks a device by itself except for plain Python values;
  * ``sync_metric_sums`` packs ALL running sums of an evaluation pass into ONE
    all_gather (the reference issues one per metric: 10 + 1 tiny collectives,
    each a full launch + rendezvous)."""
from typing import Dict, Optional, Union

import torch
import torch.distributed as dist


def _device_for_values():
    if dist.get_backend() == "nccl":
        return torch.device(f"cuda:{torch.cuda.current_device()}")
    return torch.device("cpu")


def sync_tensor_across_gpus(t: Union[torch.Tensor, None], group=None) -> Union[torch.Tensor, None]:
    """cat over ranks (dim 0) of equally shaped tensors; None passes through
    (utils_parallel.py:13-22)."""
    if t is None:
        return None
    group = dist.group.WORLD if group is None else group
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out, dim=0)


def sync_non_tensor_value_across_gpus(v, group=None) -> float:
    """sum over ranks of a Python number (utils_parallel.py:25-32; float32 like the reference)."""
    assert not torch.is_tensor(v)
    t = torch.tensor([v], dtype=torch.float, device=_device_for_values()).view(1, )
    return sync_tensor_across_gpus(t, group).sum().item()


def sync_dict_across_gpus(holder: dict, move_sync_vals_to_cpu: bool = False, group=None) -> dict:
    """{float key: 1-element tensor} merged over ranks (keys are disjoint across ranks:
    utils_parallel.py:35-64)."""
    keys = list(holder.keys())
    n = len(keys)
    assert all(isinstance(k, float) for k in keys)
    vals = [holder[k] for k in keys]
    assert all(v.numel() == 1 for v in vals)
    dev = vals[0].device if n else _device_for_values()
    kt = torch.tensor(keys, dtype=torch.float, device=dev).view(n, )
    vt = torch.stack([v.reshape(()) for v in vals]).view(n, ) if n else torch.zeros(0, device=dev)
    ks = sync_tensor_across_gpus(kt, group).cpu()
    vs = sync_tensor_across_gpus(vt, group)
    if move_sync_vals_to_cpu:
        vs = vs.cpu()
    return {k.item(): vs[i] for i, k in enumerate(ks)}


def sync_metric_sums(sums: Dict[str, torch.Tensor], count: Union[int, float],
                     group=None) -> (Dict[str, torch.Tensor], float):
    """All running metric sums of one evaluation pass + the sample count in ONE collective.
    ``sums``: name -> 0-d / 1-element tensor (same dtype & device on every rank, same keys).
    Returns (name -> sum over ranks, total count) -- what utils_trainer.py:653-674 computes
    with eleven all_gathers."""
    names = sorted(sums)
    ref = sums[names[0]]
    buf = torch.stack([sums[k].reshape(()).to(ref.dtype) for k in names]
                      + [torch.tensor(float(count), dtype=ref.dtype, device=ref.device)])
    tot = sync_tensor_across_gpus(buf.view(1, -1), group).sum(dim=0)
    return {k: tot[i] for i, k in enumerate(names)}, float(tot[-1].item())
