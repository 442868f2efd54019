from .data_sampler import DistIterSampler, per_rank_batch_size  # noqa: F401
from .pil_bicubic import make_lq_and_up, pil_bicubic_resize  # noqa: F401
