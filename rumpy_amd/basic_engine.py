"""Launch plan of the reference's "basic" models (SRCNN / VDSR, rumpy/SISR/models/basic/architectures.py:6-77) over the direct fp32
convolution kernels of the C ABI (csrc/basic_conv.hip): conv -> ReLU chains on single-channel images, optional global residual, MSE loss.

Activations are the caller-visible fp32 NCHW tensors (no layout change: these nets run on whole interpolated Y images, 1..64
channels).  Backward: the gradient at a layer's pre-activation output gives its weight / bias gradient (rumpy_dconv_wgrad) and, through the
same convolution kernel with the filter read transposed + flipped and the ReLU mask of the layer below fused into the epilogue, the gradient
at the previous layer's pre-activation output.  Parameter gradients land in the net's flat gradient buffer (views)."""
import torch

from . import _lib as L


class BasicLayer:
    def __init__(self, w, b, gw, gb):
        self.w, self.b, self.gw, self.gb = w, b, gw, gb
        self.cout, self.cin, self.k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        if w.shape[2] != w.shape[3] or self.k % 2 == 0 or self.k > 11:
            raise RuntimeError('rumpy_amd basic models: square odd kernel sizes up to 11 only (got %s)' % (tuple(w.shape),))


class BasicEngine:
    def __init__(self, layers, residual, device):
        self.lib = L.lib()
        self.layers, self.residual, self.device = layers, residual, device
        self.acts = None
        self._scratch = {}
        self._mse_partial = torch.empty(1024, dtype=torch.float32, device=device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=device)

    def repack(self, stream=None):
        """The kernels read the fp32 OIHW parameters in place: nothing to re-pack after an optimizer step."""

    def exchange_status(self):
        """No inter-workgroup exchange in these kernels (the RCAB engine's watchdog word): always clean."""
        return 0

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, x, train):
        if not x.is_cuda or x.dim() != 4 or x.shape[1] != self.layers[0].cin:
            raise RuntimeError('rumpy_amd basic models: input must be a GPU tensor [N,%d,H,W]' % self.layers[0].cin)
        x = x.float().contiguous()
        N, _, H, W = x.shape
        acts, cur, s = [x], x, self._stream()
        for i, ly in enumerate(self.layers):
            last = i == len(self.layers) - 1
            y = torch.empty(N, ly.cout, H, W, dtype=torch.float32, device=self.device)
            L.call('rumpy_dconv', L.DconvArgs(x=cur.data_ptr(), w=ly.w.data_ptr(), bias=ly.b.data_ptr(), mask=None,
                                               res=x.data_ptr() if (last and self.residual) else None, y=y.data_ptr(),
                                               N=N, Cin=ly.cin, Cout=ly.cout, H=H, W=W, k=ly.k, relu=0 if last else 1, transposed=0), s)
            acts.append(y)
            cur = y
        self.acts = acts if train else None
        return cur

    def backward(self, gout, scale=1.0, acts=None):
        """gout: gradient at the network output [N,Cout,H,W] fp32; parameter gradients (x scale) overwrite the flat gradient views.
        acts: the activation list of the forward pass this gradient belongs to (every training forward allocates its own; the autograd node
        keeps it, so two forward passes before a backward pass do not mix); default: the last training forward's."""
        acts = self.acts if acts is None else acts
        if acts is None:
            raise RuntimeError('rumpy_amd basic models: backward without a training forward pass')
        s = self._stream()
        N, _, H, W = acts[0].shape
        g = gout.float().contiguous()
        for i in range(len(self.layers) - 1, -1, -1):
            ly = self.layers[i]
            key = (N, ly.cin, ly.cout, H, W, ly.k)
            if key not in self._scratch:
                n = int(self.lib.rumpy_dconv_wgrad_partial_floats(*key))
                self._scratch[key] = torch.empty(max(1, n), dtype=torch.float32, device=self.device)
            L.call('rumpy_dconv_wgrad', L.DconvWgradArgs(x=acts[i].data_ptr(), dy=g.data_ptr(), partial=self._scratch[key].data_ptr(),
                                                         gw=ly.gw.data_ptr(), gb=ly.gb.data_ptr(), N=N, Cin=ly.cin, Cout=ly.cout,
                                                         H=H, W=W, k=ly.k, scale=float(scale)), s)
            if i > 0:
                gp = torch.empty(N, ly.cin, H, W, dtype=torch.float32, device=self.device)
                # data gradient: in-channels = this layer's outputs, out-channels = its inputs; zero where the ReLU below was off
                L.call('rumpy_dconv', L.DconvArgs(x=g.data_ptr(), w=ly.w.data_ptr(), bias=None, mask=acts[i].data_ptr(), res=None,
                                                   y=gp.data_ptr(), N=N, Cin=ly.cout, Cout=ly.cin, H=H, W=W, k=ly.k, relu=0, transposed=1), s)
                g = gp
        self.acts = None

    def mse_forward_backward(self, x, target):
        """forward + nn.MSELoss + full backward -> (loss device scalar, out)."""
        out = self.forward(x, train=True)
        target = target.float().contiguous()
        if target.shape != out.shape:
            raise RuntimeError('rumpy_amd basic models: target shape %s != output shape %s' % (tuple(target.shape), tuple(out.shape)))
        g = torch.empty_like(out)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        L.call('rumpy_mse_loss', L.MseArgs(out=out.data_ptr(), target=target.data_ptr(), grad=g.data_ptr(),
                                            partial=self._mse_partial.data_ptr(), loss=loss.data_ptr(), n=out.numel()), self._stream())
        self.backward(g)
        return loss.reshape(()), out

    def mse_eval(self, x, target):
        out = self.forward(x, train=False)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        target = target.float().contiguous()
        L.call('rumpy_mse_loss', L.MseArgs(out=out.data_ptr(), target=target.data_ptr(), grad=None,
                                            partial=self._mse_partial.data_ptr(), loss=loss.data_ptr(), n=out.numel()), self._stream())
        return out, loss.reshape(())
