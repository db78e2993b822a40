"""Kept for the tests' imports: the trajectory-sampling helpers live in the package (miccai24_immoco_amd/utils/sampling.py,
harness code over the product API only)."""
from miccai24_immoco_amd.utils.sampling import hip_psnr_samples, summarize, delta_with_se  # noqa: F401
