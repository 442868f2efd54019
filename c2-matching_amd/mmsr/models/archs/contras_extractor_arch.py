"""``ContrasExtractorSep`` (contras_extractor_arch.py:8-59): two VGG16 towers cut after conv3_1 (no ReLU -> signed
features), one for the bicubic-upsampled LR image and one for the Ref image; producer of the correlation inputs."""
import torch
import torch.nn as nn

from mmsr.models.archs.vgg_arch import NAMES, build_vgg_features


class ContrasExtractorLayer(nn.Module):

    def __init__(self):
        super().__init__()
        self.model = nn.Sequential(build_vgg_features('vgg16', NAMES['vgg16'].index('conv3_1')))
        self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, batch):
        from c2m_amd import ops as _ops
        if (not torch.is_grad_enabled() and batch.is_cuda and batch.dtype == torch.float32 and batch.shape[1] == 3 and
                (not torch.is_autocast_enabled('cuda') or _ops.bf16_autocast())):
            # inference: the five convolutions (+ ReLU) on the fused channels-last kernel; conv3_1 (no ReLU,
            # contras_extractor_arch.py:21-23) is written planar for the correlation kernels
            # (f16_range_guard: the default f16 x 2 convolution flavour covers |activation| < 65520; an input that leaves
            # it -- un-normalised 0..255 images, say -- is detected on the device and the stack recomputed on bf16 x 3)
            out = _ops.f16_range_guard(self, lambda: _ops.vgg_stack_forward(self.model._modules, batch, mean=self.mean, std=self.std,
                                                                              last_nchw=True), batch.device)
            return out['conv3_1']
        return self.model((batch - self.mean) / self.std)


class ContrasExtractorSep(nn.Module):

    def __init__(self):
        super().__init__()
        self.feature_extraction_image1 = ContrasExtractorLayer()
        self.feature_extraction_image2 = ContrasExtractorLayer()

    def forward(self, image1, image2):
        def both():
            return {'dense_features1': self.feature_extraction_image1(image1),
                    'dense_features2': self.feature_extraction_image2(image2)}
        if not torch.is_grad_enabled() and image1.is_cuda and image1.dtype == torch.float32:
            # one f16 x 2 range check (one 4-byte read-back) for both towers instead of one each
            from c2m_amd import ops as _ops
            return _ops.f16_range_guard(self, both, image1.device)
        return both()
