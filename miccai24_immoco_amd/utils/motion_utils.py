"""Line-select mask builder and motion simulator with the reference's names and semantics
(reference src/utils/motion_utils.py:7-34,56-109,121-202) on the GPU."""
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


def masks_to_col_group(masks: torch.Tensor, validate: bool = True) -> torch.Tensor:
    """[nM, H, W] one-hot column masks -> [W] int32 group index (0 = uncorrupted).

    The kernels select k-space LINES (reference immoco.py:109-111 multiplies by masks that
    extract_movement_groups builds constant down every column and disjoint between groups,
    motion_utils.py:74-107).  `validate` checks exactly that and refuses anything else instead of
    silently computing a different operator."""
    L.require_gpu(masks, what="masks_to_col_group")
    if masks.dim() != 3:
        raise L.ImmocoError("masks must be [nM, H, W]")
    m = masks.to(torch.long).contiguous()
    nM, H, W = m.shape
    if validate and nM > 0:
        ok = bool(((m == m[:, :1, :]).all() & (m[:, 0, :].sum(0) <= 1).all() & ((m == 0) | (m == 1)).all()).item())
        if not ok:
            raise L.ImmocoError("masks must be 0/1, constant down each column and disjoint between groups "
                                "(the output format of extract_movement_groups(..., make_list=True))")
    cg = torch.empty(W, device=m.device, dtype=torch.int32)
    with torch.cuda.device(m.device):
        L.check(L.lib().immoco_masks_to_groups(L.ptr(m), nM, H, W, L.ptr(cg), L.stream_ptr()), "masks_to_groups")
    return cg


# ------------------------------------------------------------------------------------------------
# motion simulator (reference motion_utils.py:7-34,121-202).  The random draws stay on the host
# torch generator in the reference's call order (a shared seed reproduces the reference's
# corruption); the image work (n affine bilinear warps with border padding, n centred FFTs, the
# band replacement) runs in HIP kernels / rocFFT.
def generate_list(size, n_movements, mingap=4, acs=24):
    """motion_utils.py:7-24 (host RNG; `acs` is unused there too)."""
    slack = size - mingap * (n_movements - 1)
    steps = torch.randint(0, slack, (1,))[0]
    inc = torch.hstack([torch.ones((steps,), dtype=torch.long), torch.zeros((n_movements,), dtype=torch.long)])
    inc = inc[torch.randperm(inc.shape[0])]
    locs = torch.argwhere(inc == 0).flatten()
    return torch.cumsum(inc, dim=0)[locs] + mingap * torch.arange(0, n_movements)


def get_rand_int(data_range, size=None):
    """motion_utils.py:27-34."""
    if size is None:
        r = torch.randint(data_range[0], data_range[1], size=(1,))
        if r == 0:
            r = r + 1
    else:
        r = torch.randint(data_range[0], data_range[1], size=size)
    return r


def motion_simulation2D(image_2d, n_movements=None):
    """Returns (ksp_corrupt [H,W] c64, mask [H,W] int64, rotations [n], translations [n,2])."""
    import ctypes as C
    from .data_utils import FFT
    L.require_gpu(image_2d, what="motion_simulation2D")
    dev = image_2d.device
    img = image_2d.to(torch.complex64).contiguous()
    H, W = img.shape
    if n_movements is None:
        n_movements = get_rand_int([5, 20]).item()
    mingap = W // n_movements
    rand_list = generate_list(W, n_movements, mingap, int(W * 0.08))
    rotations = torch.zeros((n_movements,))
    translations = torch.zeros((n_movements, 2))
    thetas, starts, ends = [], [], []
    for motion in range(n_movements):
        shift = [get_rand_int([-10, 10]).item(), get_rand_int([-10, 10]).item()]
        angle = get_rand_int([-10, 10])
        a = torch.deg2rad(angle)
        aff = torch.tensor([[torch.cos(a), -torch.sin(a), float(shift[0])],
                            [torch.sin(a), torch.cos(a), float(shift[1])]])
        aff[:, -1] /= W * 2.0 - 1          # reference: both shifts / (2*W - 1)  (motion_utils.py:163)
        thetas.append(aff)
        w0 = int(rand_list[motion])
        starts.append(w0)
        ends.append(w0 + int(get_rand_int([1, 10])))
        rotations[motion] = angle
        translations[motion, :] = torch.tensor(shift)
    theta = torch.stack(thetas).to(dev).float().contiguous()
    xs = torch.linspace(-1, 1, W, device=dev)
    ys = torch.linspace(-1, 1, H, device=dev)
    moved = torch.empty((n_movements, H, W), device=dev, dtype=torch.complex64)
    w0 = torch.tensor(starts, device=dev, dtype=torch.int32)
    w1 = torch.tensor(ends, device=dev, dtype=torch.int32)
    ksp = torch.empty((H, W), device=dev, dtype=torch.complex64)
    mask = torch.empty((H, W), device=dev, dtype=torch.long)
    with torch.cuda.device(dev):
        st = L.stream_ptr()
        L.check(L.lib().immoco_affine_warp_border(L.ptr(img), L.ptr(theta), L.ptr(xs), L.ptr(ys), n_movements, H, W,
                                                  L.ptr(moved), st), "affine_warp_border")
        k0 = FFT(img)
        kall = FFT(moved)
        L.check(L.lib().immoco_band_replace(L.ptr(k0), L.ptr(kall), L.ptr(w0), L.ptr(w1), n_movements, H, W,
                                            L.ptr(ksp), L.ptr(mask), st), "band_replace")
    return ksp, mask, rotations, translations
