"""c2m_amd -- host side of the MI355X-native C2-Matching hot path (correlation/arg-max + DCNv2 warp)."""
from . import ops  # noqa: F401
from ._lib import C2MError, LIB_PATH, device_arch, lib, profile_collect, profile_enable  # noqa: F401
