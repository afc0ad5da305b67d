"""`from mmfn_utils.models.model_img import MMFN` -> raster-map variant on the HIP engine."""
from mmfn_amd.model import MMFNImg as MMFN, PIDController  # noqa: F401
