"""Host-side counterparts of metrabs_pytorch/ptu.py that the multi-person wrapper needs.  The
tensor-heavy functions of ptu.py (softmax, soft_argmax, decode_heatmap; ptu.py:47-75) live in the HIP
decode kernel (metrabs_amd/csrc/decode.hip) and are reached through metrabs_amd.kernels."""
import torch

from metrabs_amd import kernels


def linspace(start, stop, num, dtype=None, device=None, endpoint=True):
    """ptu.linspace (ptu.py:78-92): endpoint=True with num==1 returns the MIDPOINT; endpoint=False
    stops one step short."""
    start = torch.as_tensor(start, device=device, dtype=dtype)
    stop = torch.as_tensor(stop, device=device, dtype=dtype)
    if endpoint:
        if num == 1:
            return torch.mean(torch.stack([start, stop], dim=0), dim=0, keepdim=True)
        return torch.linspace(start, stop, num, device=device, dtype=dtype)
    if num > 1:
        step = (stop - start) / num
        return torch.linspace(start, stop - step, num, device=device, dtype=dtype)
    return torch.linspace(start, stop, num, device=device, dtype=dtype)


def soft_argmax_heads(logits, n_points, config):
    """Both soft-argmaxes of MetrabsHeads.forward in one launch: ptu.soft_argmax(logits3d,
    dim=(4,3,1)) and ptu.soft_argmax(logits2d, dim=(3,2)) followed by heatmap_to_metric /
    heatmap_to_image (models/metrabs.py:78-85)."""
    return kernels.softargmax_decode(logits, n_points, config)
