"""Line-select mask builder with the reference's name and semantics
(reference src/utils/motion_utils.py:56-109): integer, bit-exact, on the GPU."""
import torch

from .. import _lib as L


def movement_col_groups(motionline_indcies: torch.Tensor):
    """[n] bool/0-1 -> (col_group [n] int32 on the GPU, number of groups)."""
    L.require_gpu(motionline_indcies, what="extract_movement_groups")
    v = motionline_indcies
    if v.dim() != 1:
        raise L.ImmocoError("extract_movement_groups expects a 1-D vector of line flags")
    # reference compares `== 1` / `== 0` (motion_utils.py:75-90)
    lines = (v == 1).to(torch.uint8).contiguous()
    n = lines.shape[0]
    cg = torch.empty(n, device=v.device, dtype=torch.int32)
    cnt = torch.empty(1, device=v.device, dtype=torch.int32)
    with torch.cuda.device(v.device):
        L.check(L.lib().immoco_extract_movement_groups(L.ptr(lines), n, L.ptr(cg), L.ptr(cnt), L.stream_ptr()),
                "extract_movement_groups")
    return cg, int(cnt.item())


def extract_movement_groups(motionline_indcies, make_list=False):
    cg, counts = movement_col_groups(motionline_indcies)
    n = cg.shape[0]
    dev = cg.device
    with torch.cuda.device(dev):
        if not make_list:
            out = torch.empty((n, n), device=dev, dtype=torch.long)
            L.check(L.lib().immoco_groups_to_matrix(L.ptr(cg), n, n, L.ptr(out), L.stream_ptr()), "groups_to_matrix")
            return out
        out = torch.empty((counts, n, n), device=dev, dtype=torch.long)
        L.check(L.lib().immoco_groups_to_masks(L.ptr(cg), counts, n, n, L.ptr(out), L.stream_ptr()), "groups_to_masks")
        return out


def masks_to_col_group(masks: torch.Tensor) -> torch.Tensor:
    """[nM, H, W] one-hot column masks -> [W] int32 group index (0 = uncorrupted)."""
    L.require_gpu(masks, what="masks_to_col_group")
    if masks.dim() != 3:
        raise L.ImmocoError("masks must be [nM, H, W]")
    m = masks.to(torch.long).contiguous()
    nM, H, W = m.shape
    cg = torch.empty(W, device=m.device, dtype=torch.int32)
    with torch.cuda.device(m.device):
        L.check(L.lib().immoco_masks_to_groups(L.ptr(m), nM, H, W, L.ptr(cg), L.stream_ptr()), "masks_to_groups")
    return cg
