from .pil_bicubic import make_lq_and_up, pil_bicubic_resize  # noqa: F401
