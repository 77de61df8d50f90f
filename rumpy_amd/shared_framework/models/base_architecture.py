"""Handler base class of the MI355X path - mirrors rumpy/shared_framework/models/base_architecture.py:17-612
(``BaseModel``): same constructor kwargs, attributes, method names, return conventions and error behaviour, so a
``<Name>Handler`` built on it drops into ``train_sisr`` / ``eval_sisr`` style callers
(rumpy/SISR/models/interface.py:97-124, rumpy/shared_framework/training/base_handler.py:219,267) unchanged.

Differences, all deliberate:
  * ``net`` computes through hand-written HIP kernels only; on a machine without the GPU/extension every compute
    call raises ``RuntimeError`` (no CPU fallback).
  * the default Adam is ``rumpy_amd.optim.FlatAdam`` (one fused kernel; identical state_dict layout).
  * with the default ``nn.L1Loss`` criterion ``run_train`` uses the fused forward+loss+backward pass; any other
    criterion / loss masking goes through the whole-network autograd node (same kernels).
  * ``set_multi_gpu`` enables one-process-per-GPU data parallelism over RCCL instead of nn.DataParallel (:70-77).
  * ``run_eval(timing=True)`` synchronises the device around the timed forward (the reference times launches, :504-508).
"""
import os
import time
from collections import OrderedDict

import numpy as np
import torch
from torch import nn, optim

from rumpy_amd.optim import FlatAdam
from rumpy_amd.SISR.models.advanced.architectures import HipSRNet


def _on_own_device(fn):
    """The HIP launches of the C ABI go to the CURRENT device; the reference selects a GPU by index (`sp_gpu`, gpu_check.py:15-25) without
    making it current.  Make the handler's device current for the duration of its compute entry points."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = self._torch_device()
        if dev.type == 'cuda' and torch.cuda.is_available():
            with torch.cuda.device(dev):
                return fn(self, *args, **kwargs)
        return fn(self, *args, **kwargs)
    return wrapped


class BaseModel(nn.Module):
    def __init__(self, device, model_save_dir, eval_mode, grad_clip=None, loss_masking=False, **kwargs):
        # unknown kwargs are accepted and ignored, like base_architecture.py:24
        super().__init__()
        self.device = torch.device('cpu') if device == 'cpu' else device
        self.criterion = nn.L1Loss()                     # :40
        self.optimizer = None
        self.net = None
        self.face_finder = False
        self.im_input = None
        self.colorspace = None
        self.steps = None
        self.eval_request_loss = True
        self.loss_masking = loss_masking
        self.grad_clip = None if grad_clip == 0 else grad_clip
        self.model_save_dir = model_save_dir
        self.eval_mode = eval_mode
        self.curr_epoch = 0
        self.state = {}
        self.learning_rate_scheduler = None
        self.legacy_load = True
        self.model_name = self.__class__.__name__.split('Handler')[0].lower()
        self.data_parallel = None
        self.defer_eval_status = False       # opt-in of a caller that checks a kept-on-device evaluation itself (run_eval)

    # ------------------------------------------------------------------ devices
    def activate_device(self):
        dev = self._torch_device()
        if dev.type == 'cuda' and torch.cuda.is_available():
            # the C-ABI launches go to torch's stream OF THIS DEVICE with raw pointers: HIP resolves a kernel for the calling thread's
            # current device, so the handler's GPU becomes the current one (one process per GPU; the reference's sp_gpu index, gpu_check.py:15-25)
            torch.cuda.set_device(dev)
        self.net.to(self.device)

    def _torch_device(self):
        if isinstance(self.device, torch.device):
            return self.device
        if isinstance(self.device, int) or (isinstance(self.device, str) and self.device.isnumeric()):
            return torch.device('cuda:%s' % self.device)
        return torch.device(self.device)

    def set_multi_gpu(self, device_ids=None):
        """Reference: nn.DataParallel(self.net) (:70-77).  Here: data parallelism is one process per GPU; gradients of
        the flat buffer are averaged over RCCL (rumpy_amd.parallel.GradientAverager) before the optimizer step."""
        from rumpy_amd.parallel import GradientAverager
        self.data_parallel = GradientAverager(self.net)
        hip = self._hip_net()
        # Early half or one all-reduce after the backward pass?  Starting the upper half's all-reduce under the remaining weight gradients costs
        # ~75 us of split launches (two half weight-gradient launches, two reductions, one more cross-stream hop) and hides half of
        # the transfer: it pays for the 62 MB of an RCAN (0.5 ms of xGMI time), not for the 6 MB of EDSR-baseline (60-100 us, half of it hidden
        # = less than the split costs).  Threshold: 4 M gradient elements; RUMPY_DP_EARLY=1 / RUMPY_DP_LATE=1 force either form (A/B).
        early = hip is not None and hip.flat_g.numel() >= (4 << 20)
        early = (early or os.environ.get('RUMPY_DP_EARLY') == '1') and os.environ.get('RUMPY_DP_LATE') != '1'
        if hip is not None and self.data_parallel.active and early:
            hip.grad_ready_hook = self.data_parallel.begin      # all-reduce of the upper half starts under the remaining weight gradients
            self.data_parallel.form = 'early'
        if self.data_parallel.active:
            if hip is not None:
                hip.data_parallel_rank = True       # a watchdog time-out raises instead of switching launch forms on one rank only (HipSRNet._watchdog_fired)
            print('Model replicated over %d GPU processes (RCCL gradient all-reduce)' % self.data_parallel.world_size)

    def set_allreduce_form(self, form):
        """Select how the gradient all-reduce of a data-parallel run is issued, after set_multi_gpu(): 'inline' = one blocking-form
        collective on the main stream behind the backward pass, 'side' = asynchronous buckets on the side stream behind the backward pass,
        'early' = the upper half of the gradient buffer on the side stream under the remaining weight gradients (two-phase weight-gradient
        plans; a later switch back keeps those plans - same gradients bit for bit, one launch more).  bench.py times 'inline' and then
        'early' inside one process group when no form is forced, so that the first run on a real node decides by itself."""
        dp, hip = self.data_parallel, self._hip_net()
        if dp is None or form not in ('inline', 'side', 'early'):
            raise RuntimeError('set_allreduce_form(%r): call set_multi_gpu() first; forms are inline | side | early' % (form,))
        if form == 'early':
            if hip is None or not hasattr(hip, 'engine_forward'):
                raise RuntimeError('the early all-reduce form needs a network on the flat-buffer engine')
            hip.grad_ready_hook = dp.begin
            dp.form = 'early'
            return
        if hip is not None and getattr(hip, 'grad_ready_hook', None) is not None:
            hip.grad_ready_hook = None
        dp.inline = form == 'inline'
        dp.form = 'inline' if (dp.inline and len(dp.buckets) == 1) else 'side'

    # ------------------------------------------------------------------ optimizer / scheduler (:79-198)
    def define_optimizer(self, optim_weights, lr=1e-4, optimizer_params=None, optimizer_type='Adam'):
        optim_weights = list(optim_weights)
        kind = optimizer_type.lower()
        if kind == 'adam':
            betas = (optimizer_params['beta_1'], optimizer_params['beta_2']) if optimizer_params is not None else (0.9, 0.999)
            hip = self._hip_net()        # the HIP network whose flat buffers the fused optimizer updates (a pipeline's generator)
            trainable = [p for p in optim_weights if p.requires_grad]
            whole_net = hip is not None and len(trainable) == len(hip.param_list) and all(a is b for a, b in zip(trainable, hip.param_list))
            if whole_net:
                return FlatAdam(hip, lr=lr, betas=betas)
            return optim.Adam([p for p in optim_weights if p.requires_grad], lr=lr, betas=betas)
        if kind == 'rmsprop':
            trainable = [p for p in optim_weights if p.requires_grad]
            if optimizer_params is not None:
                return optim.RMSprop(trainable, lr=lr, alpha=optimizer_params['alpha'])
            return optim.RMSprop(trainable, lr=lr)
        raise RuntimeError('%s optimizer not implemented' % optimizer_type)

    def define_scheduler(self, base_optimizer, scheduler, scheduler_params):
        sch = optim.lr_scheduler
        if scheduler == 'cosine_annealing_warm_restarts':
            return sch.CosineAnnealingWarmRestarts(base_optimizer, T_0=scheduler_params['restart_period'],
                                                   T_mult=scheduler_params['t_mult'], eta_min=scheduler_params['lr_min'])
        if scheduler == 'one_cycle_lr':
            return sch.OneCycleLR(base_optimizer, max_lr=scheduler_params['lr_max'], total_steps=scheduler_params['total_steps'],
                                  anneal_strategy=scheduler_params['anneal_strategy'])
        if scheduler == 'multi_step_lr':
            return sch.MultiStepLR(base_optimizer, milestones=scheduler_params['milestones'], gamma=scheduler_params['gamma'])
        if scheduler == 'step_lr':
            return sch.StepLR(base_optimizer, step_size=scheduler_params['step_size'], gamma=scheduler_params['gamma'])
        if scheduler == 'custom':
            return sch.LambdaLR(base_optimizer, lr_lambda=scheduler_params['function'])
        raise RuntimeError('%s scheduler not implemented' % scheduler)

    def training_setup(self, lr, scheduler, scheduler_params, perceptual, device, optimizer_params=None, **kwargs):
        if not self.eval_mode:
            self.optimizer = self.define_optimizer(self.net.parameters(), lr=lr, optimizer_params=optimizer_params)
            if scheduler is not None:
                self.learning_rate_scheduler = self.define_scheduler(self.optimizer, scheduler, scheduler_params)
        if perceptual is not None and self.eval_mode is False:
            raise RuntimeError('perceptual (VGG) loss is outside the MI355X hot path; train with the default L1 criterion')

    # ------------------------------------------------------------------ checkpoints (:200-394)
    @staticmethod
    def extract_model_parameters(model):
        return OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())

    def save_model(self, model_save_name, extract_state_only=False, minimal=False):
        """Layout of ``train_model_<epoch>`` as in :231-265: network / model_name / model_epoch / optimizer / scheduler_G / steps."""
        self.state['network'] = self.extract_model_parameters(self.net)
        self.state['model_name'] = self.model_name
        self.state['model_epoch'] = self.curr_epoch
        if not minimal:
            self.state['optimizer'] = self.optimizer.state_dict()
            if self.learning_rate_scheduler is not None:
                self.state['scheduler_G'] = self.learning_rate_scheduler.state_dict()
            if hasattr(self, 'steps'):
                self.state['steps'] = self.steps
        if extract_state_only:
            return self.state
        torch.save(self.state, f=os.path.join(self.model_save_dir, '{}_{}'.format(model_save_name, self.curr_epoch)))

    def load_setup(self, load_override, model_save_name, model_idx):
        if self.device == torch.device('cpu'):
            loc = self.device
        elif isinstance(self.device, int) or (isinstance(self.device, str) and self.device.isnumeric()):
            loc = 'cuda:%s' % self.device      # on PyTorch-ROCm the 'cuda' device string is the HIP device
        elif isinstance(self.device, torch.device):
            loc = self.device
        else:
            raise RuntimeError('Device %s not recognized' % self.device)
        folder = self.model_save_dir if load_override is None else load_override
        return os.path.join(folder, '{}_{}'.format(model_save_name, str(model_idx))), loc

    @staticmethod
    def legacy_switch(state_dict, qrealesrgan_fix=False):
        """Strip the historic 'model.' / 'model.module.' key prefixes (:396-412)."""
        out = OrderedDict()
        for k, v in state_dict.items():
            for prefix in ('model.module.', 'model.'):
                if k.startswith(prefix):
                    k = k[len(prefix):]
                    break
            out[k] = v
        return out

    def load_model(self, model_save_name, model_idx, legacy=False, load_override=None, preloaded_state=None,
                   config_changes=None, skip_scheduler_load=False, skip_optimizer_load=False):
        load_file, loc = self.load_setup(load_override, model_save_name, model_idx)
        state = torch.load(f=load_file, map_location=loc, weights_only=False) if preloaded_state is None else preloaded_state

        lr_key = "root['internal_params']['lr']"
        if config_changes is not None and 'values_changed' in config_changes and lr_key in config_changes['values_changed']:
            new_lr = config_changes['values_changed'][lr_key]['new_value']      # LR override through the config diff (:306-317)
            for key in [k for k in state.keys() if 'scheduler' in k.lower()]:
                state[key]['base_lrs'] = [new_lr]
                state[key]['_last_lr'] = [new_lr]
            for key in [k for k in state.keys() if 'optimizer' in k.lower()]:
                state[key]['param_groups'][0]['lr'] = new_lr

        net_state = self.legacy_switch(state['network']) if legacy else state['network']
        self.net.load_state_dict(state_dict=net_state)
        if not self.eval_mode:
            if not skip_optimizer_load and 'optimizer' in state:
                self.optimizer.load_state_dict(state['optimizer'])
            if not skip_scheduler_load and self.learning_rate_scheduler is not None and 'scheduler_G' in state:
                self.learning_rate_scheduler.load_state_dict(state['scheduler_G'])
            if hasattr(self, 'steps') and 'steps' in state:
                self.steps = state['steps']
        self.set_epoch(state['model_epoch'])
        print('Loaded model uses the following architecture:', state['model_name'])
        return state

    # ------------------------------------------------------------------ one step (:425-520)
    def standard_update(self, loss, scheduler_skip=False):
        """zero_grad / backward / [clip] / optimizer.step / scheduler.step (per batch), as :425-440."""
        self.optimizer.zero_grad()
        loss.backward()
        self._apply_update(scheduler_skip)

    def _apply_update(self, scheduler_skip=False):
        grad_mult = 1.0
        if self.data_parallel is not None:
            # RCCL all-reduce (sum) of the flat gradient buffer; the 1 / world factor travels into the fused Adam launch as grad_mult
            if isinstance(self.optimizer, FlatAdam):
                grad_mult = self.data_parallel.average_sum()
            else:
                self.data_parallel.average()            # stock torch optimizers read p.grad: the mean, in place
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.step(grad_mult=grad_mult, max_norm=self.grad_clip)
        else:
            if self.grad_clip is not None:
                nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)
            self.optimizer.step()
        if self.learning_rate_scheduler is not None and not scheduler_skip:
            self.learning_rate_scheduler.step()

    def run_model(self, x, *args, **kwargs):
        return self.net.forward(x)

    def find_loss(self, out, y):
        return self.criterion(out, y)

    def get_binary_masks(self, masks):
        new_masks = torch.zeros_like(masks)
        non_black = (masks.permute((0, 2, 3, 1)) != torch.tensor((0, 0, 0), device=masks.device)).all(-1)
        new_masks[non_black.unsqueeze(1).expand(-1, 3, -1, -1)] = 1
        return new_masks

    def _hip_net(self):
        if isinstance(self.net, HipSRNet) or (getattr(self.net, 'flat_protocol', False) and self.net.flat_p is not None):   # MoCo / a flattened Encoder
            return self.net
        gen = getattr(self.net, 'hip_generator', None)
        return gen if isinstance(gen, HipSRNet) else None

    def _check_watchdog(self):
        """generic-loss steps: the watchdog word of the step's persistent launches was staged behind the backward pass (HipSRNet.stage_step_status);
        the loss read-back above has waited for the step, so looking at it costs nothing (the fused-L1 path reads it with the loss)"""
        hip = self._hip_net()
        if hip is not None and hasattr(hip, 'check_step_status'):
            hip.check_step_status()

    def _fused_l1(self):
        hip = self._hip_net()
        return hip is not None and getattr(hip, 'supports_fused_l1', True) and type(self.criterion) is nn.L1Loss and not self.loss_masking

    def _fused_mse(self):
        """SRCNN / VDSR (basic/handlers.py:14,31: nn.MSELoss): forward + loss + backward as one pass of the direct-convolution engine"""
        return hasattr(self.net, 'fused_mse_forward_backward') and type(self.criterion) is nn.MSELoss and not self.loss_masking

    @_on_own_device
    def run_train(self, x, y, tag=None, mask=None, keep_on_device=False, scheduler_skip=False, *args, **kwargs):
        """-> (loss ndarray, out tensor (CPU unless keep_on_device)) as :457-485."""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        dev = self._torch_device()
        x, y = self._to_device(x, dev, 'x'), self._to_device(y, dev, 'y')
        host_out = False
        if self._fused_l1():
            # the reference's caller (SISRInterface.train_batch, interface.py:97-101) hands over host tensors and takes the image back on the host:
            # its device-to-host copy is queued behind the forward pass and runs under the backward pass (HipSRNet._stage_out)
            # 2 (default): copy on a copy stream, under the backward pass (1.51 ms per call); 1: on the step's own stream (2.10); 0: out.cpu() behind the
            # whole step (1.76) - same box, EDSR 32 x 48 x 48, profiles/r05_as_called.txt.  Only this mode ever creates the second stream.
            mode = os.environ.get('RUMPY_HOST_STAGING', '2')
            host_out = (not keep_on_device) and isinstance(self.net, HipSRNet) and not getattr(self.net, 'use_graph', False) and mode in ('1', '2')
            if host_out:
                loss, out = self.net.fused_l1_forward_backward(x, y, metadata=kwargs.get('extra_channels'), out_to_host='side' if mode == '2' else 'inline')
            else:
                loss, out = self.net.fused_l1_forward_backward(x, y, metadata=kwargs.get('extra_channels'))
            self._apply_update(scheduler_skip)
        elif self._fused_mse():
            loss, out = self.net.fused_mse_forward_backward(x, y)
            self._apply_update(scheduler_skip)
        else:
            out = self.run_model(x, image_names=tag, **kwargs)
            if self.loss_masking:
                binary_mask = self.get_binary_masks(mask).to(device=dev)
                out = out * binary_mask
                y = y * binary_mask
            loss = self.find_loss(out, y)
            self.standard_update(loss, scheduler_skip=scheduler_skip)
        early = self.net.take_early_loss() if hasattr(self.net, 'take_early_loss') else None   # read back after the forward pass
        loss_np = early if early is not None else loss.detach().reshape(()).cpu().numpy()
        self._check_watchdog()
        if keep_on_device:
            # in graph mode `out` is the plan's static buffer (rewritten by the next step): hand out a copy
            keep = out.detach().clone() if getattr(self.net, 'use_graph', False) else out.detach()
            return loss_np, keep
        if host_out:
            staged = self.net.take_staged_out()
            if staged is not None:
                return loss_np, staged
        return loss_np, out.detach().cpu()

    def _to_device(self, t, dev, slot):
        """a batch tensor onto the device.  Pageable host tensors (what the reference's data loader hands over) take torch's own path: measured on
        this platform (tests/tools/as_called_prof.py, profiles/r05_as_called.txt) a pageable .to(device, non_blocking=True) of the 14 MB target costs
        the host 0.15 ms, while a ring of pinned staging buffers costs 2-4 ms of cold host memcpy per step - tried and dropped."""
        return t.to(device=dev, non_blocking=True)

    @_on_own_device
    def run_eval(self, x, y=None, request_loss=False, tag=None, timing=False, keep_on_device=False, *args, **kwargs):
        """-> (out, loss | None, seconds | None) as :488-520."""
        self.net.eval()
        dev = self._torch_device()
        tic = toc = None
        hip = self._hip_net()
        if isinstance(hip, HipSRNet):
            # The status words of an evaluation pass (non-finite fp16 output, strip-exchange watchdog) are read back before the image is
            # handed out - on every return path, kept on the device or not - unless the CALLER opted in to examining them itself
            # (`defer_eval_status = True` on the handler: throughput loops that call `net.engine.check_eval()` behind their last image;
            # the words are then staged behind the pass and examined at the next one)
            hip.eval_defer = bool(self.defer_eval_status and keep_on_device and not (request_loss and y is not None) and not timing)
        with torch.no_grad():
            x = x.to(device=dev)
            want_loss = request_loss and y is not None
            if timing:
                torch.cuda.synchronize(dev)
                tic = time.perf_counter()
            if want_loss and self._fused_l1():
                out, loss_t = self.net.l1_eval(x, y.to(device=dev), metadata=kwargs.get('extra_channels'))
            elif want_loss and self._fused_mse():
                out, loss_t = self.net.mse_eval(x, y.to(device=dev))
            else:
                out = self.run_model(x, image_names=tag, **kwargs)
                loss_t = self.find_loss(out, y.to(device=dev)) if want_loss else None
            if timing:
                torch.cuda.synchronize(dev)
                toc = time.perf_counter()
            loss = loss_t.detach().reshape(()).cpu().numpy() if loss_t is not None else None
        secs = (toc - tic) if timing else None
        if isinstance(out, tuple):       # contrastive models return (embedding, q): handed back as they are (:514-515)
            return out, loss, secs
        if keep_on_device:
            return out.detach(), loss, secs
        # (the strip-exchange watchdog and the non-finite flag of an evaluation pass are checked inside SREngine.forward, on every return path)
        return out.detach().cpu(), loss, secs

    def run_forensic(self, x, *args, **kwargs):
        raise NotImplementedError('forensic dumps are outside the MI355X hot path')

    # ------------------------------------------------------------------ misc contract (:522-612)
    def print_parameters(self, verbose=False):
        total = 0
        for name, value in self.named_parameters():
            shape = getattr(value, '_rumpy_real_shape', None) or value.shape      # (a width embedded in the next kernel width counts at the reference's shape)
            if verbose:
                print(name, tuple(shape))
            total += int(np.prod(shape))
        if verbose:
            print('Total number of trainable parameters:', total)
        return total

    def print_status(self):
        raise NotImplementedError

    def epoch_end_calls(self):
        pass

    def extra_diagnostics(self):
        pass

    def pre_training_model_load(self):
        pass

    def verify_eval(self):
        return True

    def set_epoch(self, epoch):
        self.curr_epoch = epoch

    def get_learning_rate(self):
        return self.optimizer.param_groups[0]['lr']

    @staticmethod
    def best_model_selection_criteria(log_dir=None, log_file='summary.csv', model_metadata=None, stats=None,
                                      stats_dir=None, base_metric='val-PSNR'):
        """Highest value of ``base_metric`` wins (:596-612); ``stats`` is a dict of per-epoch lists."""
        if stats is None:
            raise RuntimeError('pass the loaded statistics (summary.csv reader is outside the hot path)')
        vals = list(stats[base_metric])
        lower = 'loss' in base_metric
        best = int(np.argmin(vals)) if lower else int(np.argmax(vals))
        return best if 'epoch' not in stats else int(list(stats['epoch'])[best])
