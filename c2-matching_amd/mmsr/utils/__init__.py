from .metrics import bgr2ycbcr, psnr, ssim, tensor2img_device, validation_metrics  # noqa: F401
