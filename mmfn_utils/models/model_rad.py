"""`from mmfn_utils.models.model_rad import MMFN` -> VectorNet + radar variant on the HIP engine."""
from mmfn_amd.model import MMFNRad as MMFN, PIDController  # noqa: F401
