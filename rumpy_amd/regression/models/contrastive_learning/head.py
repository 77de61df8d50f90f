"""The contrastive head of the encoder-training step on the HIP path (round 3): the mlp head of the encoder, the L2 normalisation, the MoCo /
SupMoCo logits and the softmax cross-entropy as autograd nodes over the C-ABI kernels of csrc/contrastive.hip - forward AND backward.

Reference arithmetic (paths under rumpy/regression/models/contrastive_learning/):
  encoding_models.py:43-55   ``mlp`` = Linear(256, 256), LeakyReLU(0.1, True), Linear(256, 256)
  moco.py:147-177            q = normalize(mlp(fea)); l_pos = mean_p q . k_p ; l_neg = q @ queue ; logits = cat(l_pos, l_neg) / T
  supmoco.py:75-119          l_pos = (sum_p q . k_p + q . sum_{queue entries of q's class} key) / T / (P + their number)
  handlers.py:57             nn.CrossEntropyLoss(logits, zeros)
Round 2 ran these as torch ops (rocBLAS GEMMs + ATen kernels + torch autograd); everything here is exact fp32 like the reference.
No CPU path: tensors must be on the GPU."""
import torch

from rumpy_amd import _lib as L

SLOPE = 0.1


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('rumpy_amd: the contrastive head runs on an MI355X through the HIP extension; there is no CPU path')


def sgemm(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha=1.0, bias=None, slope=1.0, accumulate=False, a_off=0, c_off=0):
    """C[m, n] = act(alpha * sum_k A(m, k) B(k, n) + bias[n]) [+ C] through rumpy_sgemm (element offsets a_off / c_off into A / C)"""
    pf = int(L.lib().rumpy_sgemm_partial_floats(M, N, K))
    part = torch.empty(pf, dtype=torch.float32, device=C.device) if pf else None
    L.call('rumpy_sgemm', L.SgemmArgs(A=A.data_ptr() + 4 * a_off, B=B.data_ptr(), C=C.data_ptr() + 4 * c_off, bias=_p(bias), partial=_p(part), M=M, N=N, K=K, ldc=ldc,
                                      sam=sam, sak=sak, sbk=sbk, sbn=sbn, alpha=float(alpha), leaky_slope=float(slope), accumulate=1 if accumulate else 0),
           _stream(C))


def _grad_into(p, compute):
    """run compute(dst, accumulate) for a parameter's gradient: straight into ``p.grad`` when it exists as a contiguous fp32 tensor (the flat
    gradient buffer's views of the encoders: no torch add), else into a fresh tensor that autograd accumulates"""
    g = p.grad
    if g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda:
        compute(g, True)
        return None
    out = torch.empty_like(p, dtype=torch.float32)
    compute(out, False)
    return out


class _MlpHeadFn(torch.autograd.Function):
    """q = W2 lrelu(W1 fea + b1) + b2 on [N, C] rows"""

    @staticmethod
    def forward(ctx, fea, w1, b1, w2, b2):
        _need_gpu(fea, w1, w2)
        fea = fea.contiguous().float()
        n, c = fea.shape
        hdim, o = w1.shape[0], w2.shape[0]
        h = torch.empty(n, hdim, dtype=torch.float32, device=fea.device)
        q = torch.empty(n, o, dtype=torch.float32, device=fea.device)
        sgemm(fea, w1, h, n, hdim, c, c, 1, 1, c, hdim, bias=b1, slope=SLOPE)            # B(k, j) = w1[j, k]
        sgemm(h, w2, q, n, o, hdim, hdim, 1, 1, hdim, o, bias=b2)
        ctx.save_for_backward(fea, h, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        return q

    @staticmethod
    def backward(ctx, dq):
        fea, h, w1d, w2d = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        dq = dq.contiguous().float()
        n, c = fea.shape
        hdim, o = w1d.shape[0], w2d.shape[0]
        s = _stream(dq)
        lib = L.lib()
        need = ctx.needs_input_grad
        gw2 = gb2 = gw1 = gb1 = dfea = None
        if need[3]:     # dW2[o, i] = sum_n dq[n, o] h[n, i]
            gw2 = _grad_into(w2, lambda dst, acc: sgemm(dq, h, dst, o, hdim, n, 1, o, hdim, 1, hdim, accumulate=acc))
        if need[4]:
            def colsum_b2(dst, acc):
                if acc:
                    tmp = torch.empty(o, dtype=torch.float32, device=dq.device)
                    L.check(lib.rumpy_colsum(dq.data_ptr(), tmp.data_ptr(), n, o, s), 'rumpy_colsum')
                    L.check(lib.rumpy_row_axpy(dst.data_ptr(), tmp.data_ptr(), _ones(dq.device).data_ptr(), 1, o, 1, s), 'rumpy_row_axpy')
                else:
                    L.check(lib.rumpy_colsum(dq.data_ptr(), dst.data_ptr(), n, o, s), 'rumpy_colsum')
            gb2 = _grad_into(b2, colsum_b2)
        if need[0] or need[1] or need[2]:
            dh = torch.empty(n, hdim, dtype=torch.float32, device=dq.device)
            sgemm(dq, w2d, dh, n, hdim, o, o, 1, hdim, 1, hdim)                          # dh = dq W2 : B(k = o, j = i) = w2[o, i]
            L.check(lib.rumpy_lrelu_bwd(dh.data_ptr(), h.data_ptr(), dh.numel(), SLOPE, s), 'rumpy_lrelu_bwd')
            if need[1]:
                gw1 = _grad_into(w1, lambda dst, acc: sgemm(dh, fea, dst, hdim, c, n, 1, hdim, c, 1, c, accumulate=acc))
            if need[2]:
                def colsum_b1(dst, acc):
                    if acc:
                        tmp = torch.empty(hdim, dtype=torch.float32, device=dq.device)
                        L.check(lib.rumpy_colsum(dh.data_ptr(), tmp.data_ptr(), n, hdim, s), 'rumpy_colsum')
                        L.check(lib.rumpy_row_axpy(dst.data_ptr(), tmp.data_ptr(), _ones(dq.device).data_ptr(), 1, hdim, 1, s), 'rumpy_row_axpy')
                    else:
                        L.check(lib.rumpy_colsum(dh.data_ptr(), dst.data_ptr(), n, hdim, s), 'rumpy_colsum')
                gb1 = _grad_into(b1, colsum_b1)
            if need[0]:
                dfea = torch.empty(n, c, dtype=torch.float32, device=dq.device)
                sgemm(dh, w1d, dfea, n, c, hdim, hdim, 1, c, 1, c)                        # dfea = dh W1
        return dfea, gw1, gb1, gw2, gb2


_ONES = {}


def _ones(dev):
    t = _ONES.get(dev)
    if t is None:
        t = _ONES[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    return t


def mlp_head(mlp, fea):
    """the encoder's ``mlp`` Sequential (Linear, LeakyReLU, Linear) on fea [N, C]"""
    lin1, lin2 = mlp[0], mlp[2]
    return _MlpHeadFn.apply(fea, lin1.weight, lin1.bias, lin2.weight, lin2.bias)


def normalize_rows(x):
    """nn.functional.normalize(x, dim=1) without a gradient path (the keys)"""
    _need_gpu(x)
    x = x.detach().contiguous().float()
    y = torch.empty_like(x)
    L.check(L.lib().rumpy_l2norm_rows(x.data_ptr(), y.data_ptr(), None, x.shape[0], x.shape[1], _stream(x)), 'rumpy_l2norm_rows')
    return y


class _NormLogitsFn(torch.autograd.Function):
    """logits[n, 0] = qn[n] . v[n] ; logits[n, 1 + j] = qn[n] . queue[:, j] / T with qn = normalize(q); v and the queue carry no gradient"""

    @staticmethod
    def forward(ctx, q, v, queue, inv_t):
        _need_gpu(q, v, queue)
        q = q.contiguous().float()
        n, c = q.shape
        k = queue.shape[1]
        lib, s = L.lib(), _stream(q)
        qn = torch.empty_like(q)
        inv = torch.empty(n, dtype=torch.float32, device=q.device)
        L.check(lib.rumpy_l2norm_rows(q.data_ptr(), qn.data_ptr(), inv.data_ptr(), n, c, s), 'rumpy_l2norm_rows')
        logits = torch.empty(n, 1 + k, dtype=torch.float32, device=q.device)
        L.check(lib.rumpy_rowdot(qn.data_ptr(), v.data_ptr(), logits.data_ptr(), n, c, 1 + k, 1.0, s), 'rumpy_rowdot')
        sgemm(qn, queue, logits, n, k, c, c, 1, k, 1, 1 + k, alpha=inv_t, c_off=1)
        ctx.save_for_backward(qn, inv, v, queue)
        ctx.inv_t = inv_t
        return logits

    @staticmethod
    def backward(ctx, dl):
        qn, inv, v, queue = ctx.saved_tensors
        dl = dl.contiguous().float()
        n, c = qn.shape
        k = queue.shape[1]
        lib, s = L.lib(), _stream(dl)
        dqn = torch.empty_like(qn)
        sgemm(dl, queue, dqn, n, c, k, 1 + k, 1, 1, k, c, alpha=ctx.inv_t, a_off=1)      # dl[:, 1:] @ queue^T / T : B(k = j, i = c) = queue[c, j]
        L.check(lib.rumpy_row_axpy(dqn.data_ptr(), v.data_ptr(), dl.data_ptr(), n, c, 1 + k, s), 'rumpy_row_axpy')      # + dl[:, 0] * v
        dq = torch.empty_like(qn)
        L.check(lib.rumpy_l2norm_rows_bwd(dqn.data_ptr(), qn.data_ptr(), inv.data_ptr(), dq.data_ptr(), n, c, s), 'rumpy_l2norm_rows_bwd')
        return dq, None, None, None


def moco_logits(q, k, queue, temperature, positives, labels=None, queue_labels=None):
    """q [N, C] (mlp output, with gradient), k [N * positives, C] normalised keys, queue [C, K] -> logits [N, 1 + K].
    labels / queue_labels (int64): SupMoCo's positives from the queue."""
    _need_gpu(q, k, queue)
    n, c = q.shape
    kq = queue.shape[1]
    lib, s = L.lib(), _stream(q)
    k = k.detach().contiguous().float()
    qc = queue.detach().clone()         # the enqueue that follows rewrites columns of the queue before the backward pass reads it (moco.py:172)
    v = torch.empty(n, c, dtype=torch.float32, device=q.device)
    if labels is None:
        L.check(lib.rumpy_pos_vector(k.data_ptr(), None, None, v.data_ptr(), n, positives, c, 1.0 / temperature, s), 'rumpy_pos_vector')
    else:
        same = torch.empty(n, kq, dtype=torch.float32, device=q.device)
        cnt = torch.empty(n, dtype=torch.float32, device=q.device)
        L.check(lib.rumpy_label_match(labels.data_ptr(), queue_labels.data_ptr(), same.data_ptr(), cnt.data_ptr(), n, kq, s), 'rumpy_label_match')
        ssum = torch.empty(n, c, dtype=torch.float32, device=q.device)
        sgemm(same, qc, ssum, n, c, kq, kq, 1, 1, kq, c)                                  # same @ queue^T
        L.check(lib.rumpy_pos_vector(k.data_ptr(), ssum.data_ptr(), cnt.data_ptr(), v.data_ptr(), n, positives, c, 1.0 / temperature, s), 'rumpy_pos_vector')
    return _NormLogitsFn.apply(q, v, qc, 1.0 / temperature)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        _need_gpu(logits, target)
        logits = logits.contiguous().float()
        target = target.to(device=logits.device, dtype=torch.int64).contiguous()
        n, m = logits.shape
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        rowloss = torch.empty(n, dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        L.check(L.lib().rumpy_ce_rows(logits.data_ptr(), target.data_ptr(), lse.data_ptr(), rowloss.data_ptr(), loss.data_ptr(), n, m, _stream(logits)), 'rumpy_ce_rows')
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        n, m = logits.shape
        g = g.contiguous().float()
        dl = torch.empty_like(logits)
        L.check(L.lib().rumpy_ce_rows_bwd(logits.data_ptr(), target.data_ptr(), lse.data_ptr(), g.data_ptr(), dl.data_ptr(), n, m, _stream(logits)), 'rumpy_ce_rows_bwd')
        return dl, None


class HipCrossEntropyLoss(torch.nn.Module):
    """nn.CrossEntropyLoss() (mean reduction, integer class targets) on [N, M] fp32 logits: forward and backward in HIP"""

    def forward(self, logits, target):
        if logits.dim() != 2:
            raise RuntimeError('rumpy_amd: HipCrossEntropyLoss takes [N, M] logits')
        return _CrossEntropyFn.apply(logits, target)
