"""Stage-3 (MSE yaml) slice of the reference's ``RefRestorationModel`` (ref_restoration_model.py:21-43, 47-87,
186-279): net construction through the registry, the four Adam parameter groups, ``feed_data`` / ``optimize_parameters``
with the pixel loss / ``test``.  GAN, perceptual, texture losses, validation image I/O and LR schedulers are outside the
hot path and are not provided here -- ``mmsr/train.py`` of the reference keeps using its own
model file; this class is what bench/tests drive.

Validation: ``nondist_validation`` computes PSNR / PSNR_Y / SSIM_Y with the reference's rules
(ref_restoration_model.py:295-370) but on the device (mmsr/utils/metrics.py here: three scalars per image cross PCIe
instead of two full images); ``dist_validation`` -- which in the reference calls a method that does not exist
(sr_model.py:160-162) -- shards the loader over the ranks and all-reduces the sums.

MI355X addition -- ``train: {hip_graph: true}`` (single-process training): the whole training step (extractor +
correspondence + net_g forward + L1 + backward + Adam) is a launch-bound chain of ~1500 small kernels at the reference's
stage-3 crop size (GT 160x160, batch 4 per GPU): the host spends more time launching than the GPU computing.  The step is
captured once into a hipGraph (torch.cuda.CUDAGraph: static input buffers filled by ``feed_data``, Adam in capturable
mode, two eager warm-up steps so that allocator, MIOpen find and lazy initialisation stay outside the capture) and
replayed per ``optimize_parameters`` call -- same arithmetic, same kernels, one launch.
"""
import copy
import logging

import torch

import mmsr.models.networks as networks
from mmsr.models.base_model import BaseModel, unwrap
from mmsr.utils import metrics

logger = logging.getLogger('base')


class RefRestorationModel(BaseModel):

    def __init__(self, opt):
        super().__init__(opt)
        opt = copy.deepcopy(opt)  # the factories pop 'type'
        self.net_g = self.model_to_device(networks.define_net_g(opt))
        # net_map has no trainable parameters; net_extractor's never receive gradients in stage 3: both stay bare
        self.net_map = self.model_to_device(networks.define_net_map(opt), receives_gradients=False)
        self.net_extractor = self.model_to_device(networks.define_net_extractor(opt), receives_gradients=False)
        for p in self.net_extractor.parameters():
            p.requires_grad = False
        path = self.opt.get('path') or {}
        if path.get('pretrain_model_feature_extractor'):
            self.load_network(self.net_extractor, path['pretrain_model_feature_extractor'], path.get('strict_load', True))
        if path.get('pretrain_model_g'):
            self.load_network(self.net_g, path['pretrain_model_g'], path.get('strict_load', True))
        # (one process, ONE device: nn.DataParallel's scatter / replicate / worker threads cannot be captured)
        self._graph_on = bool(self.is_train and (self.opt.get('train') or {}).get('hip_graph') and not self.opt.get('dist')
                              and self.device.type == 'cuda' and not isinstance(self.net_g, torch.nn.DataParallel))
        if self._graph_on:
            # the captured step cannot read the f16 x 2 range flag back (c2m_amd.ops.f16_range_guard): the frozen nets that run
            # under no_grad inside it keep the full-range bf16 x 3 flavour from the first (eager) step on
            for net in (self.net_extractor, self.net_map):
                for m in net.modules():
                    m._c2m_conv_bf16x3 = True
        self._eval_feed = False   # True while a validation loop feeds data: its batches bypass the graph's static buffers
        self._graph, self._graph_calls, self._static = None, 0, None
        if self.is_train:
            self.net_g.train()
            self._build_optimizer()
            self.cri_pix = torch.nn.L1Loss()  # pixel_criterion: L1Loss, pixel_weight 1.0 (stage3_restoration_mse.yml:84-85)
            self.pixel_weight = float(self.opt['train'].get('pixel_weight', 1.0))

    def _build_optimizer(self):
        """Adam with the reference's four groups keyed on parameter names (ref_restoration_model.py:47-87)."""
        t = self.opt['train']
        groups = {'g': [], 'offset': [], 'relu3_offset': [], 'relu2_offset': []}
        for name, p in unwrap(self.net_g).named_parameters():
            if not p.requires_grad:
                continue
            if 'offset' in name:
                key = 'relu3_offset' if 'small' in name else 'relu2_offset' if 'medium' in name else 'offset'
            else:
                key = 'g'
            groups[key].append(p)
        self.optimizer_g = torch.optim.Adam(
            [{'params': groups['g']},
             {'params': groups['offset'], 'lr': t['lr_offset']},
             {'params': groups['relu3_offset'], 'lr': t['lr_relu3_offset']},
             {'params': groups['relu2_offset'], 'lr': t['lr_relu2_offset']}],
            lr=t['lr_g'], weight_decay=t.get('weight_decay_g', 0), betas=tuple(t['beta_g']),
            capturable=self._graph_on)   # (step counters on the device: required inside a captured step)
        self.optimizers.append(self.optimizer_g)

    _FEED = (('img_in_lq', 'img_in_lq'), ('img_ref', 'img_ref'), ('gt', 'img_in'), ('match_img_in', 'img_in_up'))

    def feed_data(self, data):
        if self._graph_on and not self._eval_feed:
            # static input buffers: the captured step reads these addresses; a new batch is copied INTO them
            shapes = {a: tuple(data[k].shape) for a, k in self._FEED}
            if self._static is None or {a: tuple(t.shape) for a, t in self._static.items()} != shapes:
                self._static = {a: data[k].to(self.device, copy=True) for a, k in self._FEED}
                self._graph, self._graph_calls = None, 0    # new geometry: capture again
            else:
                for a, k in self._FEED:
                    self._static[a].copy_(data[k], non_blocking=True)
            for a, _ in self._FEED:
                setattr(self, a, self._static[a])
            return
        self.img_in_lq = data['img_in_lq'].to(self.device)
        self.img_ref = data['img_ref'].to(self.device)
        self.gt = data['img_in'].to(self.device)
        self.match_img_in = data['img_in_up'].to(self.device)

    def _correspondence(self):
        with torch.no_grad():  # non-differentiable past the arg-max; keeps DDP's reducer to net_g only
            self.features = self.net_extractor(self.match_img_in, self.img_ref)
            self.pre_offset, self.img_ref_feat = self.net_map(self.features, self.img_ref)

    def optimize_parameters(self, step):
        if self._graph_on:
            return self._optimize_graphed()
        self._train_step()

    def _train_step(self):
        self._correspondence()
        self.output = self.net_g(self.img_in_lq, self.pre_offset, self.img_ref_feat)
        self.optimizer_g.zero_grad()
        l_pix = self.cri_pix(self.output, self.gt) * self.pixel_weight
        l_pix.backward()  # DCNv2 backward x3; DDP all-reduces net_g's gradients (RCCL) while it runs
        self.optimizer_g.step()
        self.log_dict['l_g_pix'] = l_pix.detach()  # no .item(): the reference's per-step host sync is dropped

    GRAPH_WARMUP_STEPS = 2

    def _optimize_graphed(self):
        """Every call is exactly one training step: the first GRAPH_WARMUP_STEPS run eagerly on a side stream, the next one
        captures the step and replays it once, later ones only replay.  log_dict / output / the parameters' .grad keep
        their addresses: each replay rewrites them in place."""
        if self._graph is not None:
            self._graph.replay()
            self.output = self._graph_output   # (a validation in between re-pointed / deleted the attribute: ADVICE r4)
            return
        self._graph_calls += 1
        cur = torch.cuda.current_stream()
        if self._graph_calls <= self.GRAPH_WARMUP_STEPS:
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._train_step()
            cur.wait_stream(side)
            return
        self.optimizer_g.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._train_step()
        self._graph = graph
        self._graph_output = self.output       # the tensor every replay rewrites in place
        graph.replay()

    # ---------------------------------------------------------------- validation
    def _validate_shard(self, dataloader, rank, world):
        """-> float64 tensor [psnr_sum, psnr_y_sum, ssim_y_sum, count] over items idx % world == rank."""
        crop = self.opt.get('crop_border')
        crop = self.opt.get('scale', 4) if crop is None else crop   # options.py:56-57
        sums = torch.zeros(4, dtype=torch.float64, device=self.device)
        for idx, val_data in enumerate(dataloader):
            if idx % world != rank:
                continue
            self._eval_feed = True    # (differently shaped validation pairs must not reset the captured training step)
            try:
                self.feed_data(val_data)
            finally:
                self._eval_feed = False
            sr = self.test()
            gt = self.gt
            if val_data.get('padding', False) is not False and bool(torch.as_tensor(val_data['padding']).any()):
                oh, ow = (int(torch.as_tensor(v).flatten()[0]) for v in val_data['original_size'][:2])
                sr = sr[..., :oh, :ow]          # the reference crops only the SR image (:311-315); shapes must then agree
                gt = gt[..., :oh, :ow]
            m = metrics.validation_metrics(sr, gt, crop_border=crop)
            sums[0] += m['psnr'].sum()
            sums[1] += m['psnr_y'].sum()
            sums[2] += m['ssim_y'].sum()
            sums[3] += sr.shape[0]
            del self.img_in_lq, self.output, self.gt   # as the reference: keep peak memory at one batch (:331-335)
        return sums

    def _report(self, sums, name, current_iter, tb_logger):
        n = max(float(sums[3]), 1.0)
        res = {'psnr': float(sums[0]) / n, 'psnr_y': float(sums[1]) / n, 'ssim_y': float(sums[2]) / n, 'count': int(sums[3])}
        if self.rank <= 0:
            logger.info(f"# Validation {name} # PSNR: {res['psnr']:.4e} # PSNR_Y: {res['psnr_y']:.4e} "
                        f"# SSIM_Y: {res['ssim_y']:.4e}.")
            if tb_logger:
                for k in ('psnr', 'psnr_y', 'ssim_y'):
                    tb_logger.add_scalar(k, res[k], current_iter)
        return res

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img):
        if save_img:
            raise NotImplementedError('image writing is outside the hot path (SURVEY.md 2.1)')
        name = getattr(getattr(dataloader, 'dataset', None), 'opt', {}).get('name', 'val')
        return self._report(self._validate_shard(dataloader, 0, 1), name, current_iter, tb_logger)

    def dist_validation(self, dataloader, current_iter, tb_logger, save_img):
        """Every rank validates its share of the loader; one all-reduce (RCCL on the GPU, gloo in the CPU tests) of four
        float64 sums gives every rank the dataset averages."""
        if save_img:
            raise NotImplementedError('image writing is outside the hot path (SURVEY.md 2.1)')
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        sums = self._validate_shard(dataloader, rank, world)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        name = getattr(getattr(dataloader, 'dataset', None), 'opt', {}).get('name', 'val')
        return self._report(sums, name, current_iter, tb_logger)

    def test(self):
        self.net_g.eval()
        with torch.no_grad():
            def whole():
                self._correspondence()
                return self.net_g(self.img_in_lq, self.pre_offset, self.img_ref_feat)
            if self.img_in_lq.is_cuda:
                # ONE f16 x 2 range check for the whole inference step (extractor, VGG taps, RestorationNet): the module
                # forwards' own guards nest inside it and skip their read-backs -- one 4-byte device-to-host sync per step
                # instead of three (each one drains the GPU queue: ~0.3 ms of a 139 ms configs[2] step).  On overflow the whole
                # step is recomputed on the full-range flavour and this model keeps it (ops.f16_range_guard).
                from c2m_amd import ops as _ops
                self.output = _ops.f16_range_guard(self, whole, self.img_in_lq.device)
            else:
                self.output = whole()
        if self.is_train:
            self.net_g.train()
        return self.output
