from mmfn_amd.config import GlobalConfig  # noqa: F401
