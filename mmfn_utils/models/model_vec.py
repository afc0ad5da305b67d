"""`from mmfn_utils.models.model_vec import MMFN` -> VectorNet-map variant on the HIP engine."""
from mmfn_amd.model import MMFN, PIDController  # noqa: F401
