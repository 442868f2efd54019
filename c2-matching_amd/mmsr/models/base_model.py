"""Device placement and data-parallel wrapping for the MI355X path: the changed ``BaseModel.model_to_device``
(reference: mmsr/models/base_model.py:62-75) plus the save/load helpers that depend on the wrapping (:185-265) and the
training-state save / resume pair (:267-307), so that the file can replace the reference's wholesale.

Why it changed ("DDP on modern PyTorch"): the reference wraps EVERY net in DistributedDataParallel.
With the installed PyTorch that raises for ``net_map`` (all parameters frozen) and leaves ``net_extractor`` (trainable
parameters that never see a backward) with unfinished reductions.  Here only nets whose parameters actually receive
gradients are wrapped; everything else stays a bare module and runs under ``no_grad`` (the arg-max cuts the graph
anyway).  One process per GPU; ``backend='nccl'`` of PyTorch-ROCm IS RCCL, the 35.5 MB gradient all-reduce of ``net_g``
rides the xGMI mesh in DDP's buckets, overlapped with backward.
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn.parallel import DataParallel, DistributedDataParallel

logger = logging.getLogger('base')


def unwrap(net):
    return net.module if isinstance(net, (DataParallel, DistributedDataParallel)) else net


class BaseModel:

    def __init__(self, opt):
        self.opt = opt
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank = torch.distributed.get_rank()
        else:
            self.rank = -1
        self.device = torch.device('cuda' if opt.get('gpu_ids') is not None else 'cpu')
        self.is_train = opt.get('is_train', False)
        self.schedulers = []
        self.optimizers = []
        self.log_dict = OrderedDict()

    def model_to_device(self, net, receives_gradients=None):
        """Move to the device and wrap for data parallelism.

        receives_gradients: None = infer (any parameter with requires_grad); False = never wrap (frozen nets and nets
        that only run under no_grad, e.g. net_map / net_extractor in stage 3)."""
        net = net.to(self.device)
        if receives_gradients is None:
            receives_gradients = any(p.requires_grad for p in net.parameters())
        if self.opt.get('dist'):
            if not receives_gradients:
                return net
            if self.device.type == 'cuda':
                return DistributedDataParallel(net, device_ids=[torch.cuda.current_device()],
                                               gradient_as_bucket_view=True, broadcast_buffers=False)
            return DistributedDataParallel(net)  # gloo / CPU: used by the world_size-2 tests
        gpu_ids = self.opt.get('gpu_ids') or []
        if self.device.type == 'cuda' and len(gpu_ids) > 1:
            return DataParallel(net)  # reference default without a launcher (base_model.py:73-74)
        return net

    def validation(self, dataloader, current_iter, tb_logger, save_img=False):
        """Dispatch as the reference does (base_model.py:44-58)."""
        if self.opt.get('dist'):
            return self.dist_validation(dataloader, current_iter, tb_logger, save_img)
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def get_bare_model(self, net):
        return unwrap(net)

    def get_current_log(self):
        return self.log_dict

    def save_network(self, net, net_label, current_iter):
        """state_dict without the wrapper's `module.` prefix, tensors on the CPU (base_model.py:185-206)."""
        if self.rank > 0:
            return
        name = f'{net_label}_latest.pth' if current_iter == -1 else f'{net_label}_{current_iter}.pth'
        path = os.path.join(self.opt['path']['models'], name)
        state = OrderedDict((k[7:] if k.startswith('module.') else k, v.cpu()) for k, v in unwrap(net).state_dict().items())
        torch.save(state, path)

    def load_network(self, net, load_path, strict=True):
        """Accepts checkpoints saved with or without the `module.` prefix (base_model.py:208-265)."""
        net = unwrap(net)
        logger.info(f'Loading {net.__class__.__name__} model from {load_path}.')
        raw = torch.load(load_path, map_location='cpu')
        state = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in raw.items())
        net.load_state_dict(state, strict=strict)
        from c2m_amd import ops as _ops
        _ops.clear_weight_caches()   # (load_state_dict bumps the version counters; this also frees the old images)

    def save_training_state(self, epoch, current_iter):
        """Optimizer / scheduler states for resuming, written by the master rank only, as `<iter>.state` under
        opt['path']['training_state'] with the reference's keys (base_model.py:267-290); nothing for current_iter == -1."""
        if self.rank > 0 or current_iter == -1:
            return
        state = {'epoch': epoch, 'iter': current_iter,
                 'optimizers': [o.state_dict() for o in self.optimizers],
                 'schedulers': [s.state_dict() for s in self.schedulers]}
        torch.save(state, os.path.join(self.opt['path']['training_state'], f'{current_iter}.state'))

    def resume_training(self, resume_state):
        """Reload the optimizers and schedulers from a `.state` dict (base_model.py:292-307; same length checks)."""
        resume_optimizers, resume_schedulers = resume_state['optimizers'], resume_state['schedulers']
        assert len(resume_optimizers) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(resume_schedulers) == len(self.schedulers), 'Wrong lengths of schedulers'
        for o, st in zip(self.optimizers, resume_optimizers):
            o.load_state_dict(st)
        for sc, st in zip(self.schedulers, resume_schedulers):
            sc.load_state_dict(st)
