from .corr import CorrBlock, bilinear_sampler, coords_grid, upflow8  # noqa: F401
from .extractor import BasicEncoder, SmallEncoder  # noqa: F401
from .update import BasicUpdateBlock, SmallUpdateBlock  # noqa: F401
