"""SRCNN / VDSR of the MI355X path: parameter containers with the reference's state_dict keys (``layer_dict.conv_<i>.weight|bias``,
rumpy/SISR/models/basic/architectures.py:6-77) over the direct fp32 convolution kernels (rumpy_amd/basic_engine.py).
BASELINE config 0 (SRCNN x2 on the example data) is the reference's own CPU-runnable case; here it runs on the GPU through the same
handler API, with exact fp32 arithmetic (these layers - 9x9 on one channel, one output channel - are not MFMA-shaped)."""
import torch
from torch import nn

from rumpy_amd.basic_engine import BasicEngine, BasicLayer
from rumpy_amd.SISR.models.advanced.architectures import HipSRNet


class _BasicFn(torch.autograd.Function):
    """Whole-network autograd node for criteria other than the stock nn.MSELoss: both directions run the HIP kernels."""

    @staticmethod
    def forward(ctx, x, net, *params):
        ctx.net = net
        out = net.engine.forward(x, train=True)
        ctx.acts = net.engine.acts             # this pass's own activations: a second forward before backward() does not replace them
        return out

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        net.engine.backward(gout, acts=ctx.acts)
        net.attach_grads()
        return (None, None) + tuple(None for _ in net.param_list)


class SRCNN(HipSRNet):
    residual = False
    supports_fused_l1 = False          # BaseModel: the fused L1 pass belongs to the EDSR / RCAN engine

    def __init__(self, kernel_pattern=None, channel_pattern=None, padding='same'):
        super().__init__()
        kernel_pattern = [9, 5, 5] if kernel_pattern is None else list(kernel_pattern)                  # :17-20
        channel_pattern = [1, 64, 32, 1] if channel_pattern is None else list(channel_pattern)
        if padding != 'same':
            raise NotImplementedError('rumpy_amd basic models: only padding="same" is built (the reference default, architectures.py:26-29)')
        if len(channel_pattern) != len(kernel_pattern) + 1:
            raise ValueError('channel_pattern needs one more entry than kernel_pattern')
        self.layer_dict = nn.ModuleDict()
        self.depth = len(kernel_pattern)
        for i, k in enumerate(kernel_pattern):
            self.layer_dict['conv_{}'.format(i)] = nn.Conv2d(channel_pattern[i], channel_pattern[i + 1], kernel_size=k, padding=k // 2)
        self._finalize()

    # ---- engine ----
    def _ensure_engine(self):
        if not self.flat_p.is_cuda:
            raise RuntimeError('rumpy_amd: this network only runs on an MI355X through the HIP extension; '
                               'there is no CPU path (parameters are on %s)' % self.flat_p.device)
        if self.engine is None:
            idx = {id(p): i for i, p in enumerate(self.param_list)}
            layers = []
            for i in range(self.depth):
                m = self.layer_dict['conv_{}'.format(i)]
                layers.append(BasicLayer(m.weight.data, m.bias.data, self.grad_views[idx[id(m.weight)]], self.grad_views[idx[id(m.bias)]]))
            self.engine = BasicEngine(layers, self.residual, self.flat_p.device)

    def mark_weights_updated(self):
        self._ensure_engine()

    def forward(self, x, metadata=None):
        self._ensure_engine()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.param_list):
            return _BasicFn.apply(x, self, *self.param_list)
        return self.engine.forward(x, train=False)

    def fused_mse_forward_backward(self, x, y):
        """forward + nn.MSELoss + full backward in one pass (base_architecture.py:474-480 minus the optimizer) -> (loss, out)."""
        self._ensure_engine()
        loss, out = self.engine.mse_forward_backward(x, y)
        self._stage_loss(loss)
        return loss, out

    def mse_eval(self, x, y):
        self._ensure_engine()
        return self.engine.mse_eval(x, y)

    def fused_l1_forward_backward(self, x, y, metadata=None):
        raise RuntimeError('rumpy_amd basic models train with nn.MSELoss (basic/handlers.py:14)')

    def reset_parameters(self):
        """architectures.py:54-60"""
        for layer in self.layer_dict.children():
            layer.reset_parameters()


class VDSR(SRCNN):
    """Deeper SRCNN whose output is added to its input (architectures.py:63-77)."""
    residual = True
