"""CPU oracle for the RUMpy convolutional SR hot path (EDSR / RCAN train + eval step).

TEST INFRASTRUCTURE ONLY.  Nothing under ``rumpy_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
only as the checker / the reported CPU baseline - never as the thing shipped.

What it is: a plain PyTorch-CPU fp32 restatement of the reference arithmetic.  The reference
(um-dsrg/RUMpy v1.0) is 100 % Python on PyTorch, so the arithmetic itself lives in the
third-party dependency ``pytorch>=1.10.0`` (unpinned, reference ``requirements.txt:1``); the
restatement therefore uses the same ATen CPU ops (conv2d, pixel_shuffle, adaptive mean,
sigmoid, l1, Adam, cosine warm restarts) wired exactly like the reference modules.

Parity pinning: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` in the build container (torch 2.10.0 CPU) and stores inputs/outputs of
its own modules and handlers as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function below against them (bit-exact for forward, <=1e-6 for the train step).
The reference's own tests pin shapes only (automated_testing/sisr_tests/test_model_cpu_execute.py:33-49)
plus the parameter counts in rumpy/sr_tools/stats.py:238, which are asserted too.

Every builder cites the reference file:line it restates (paths relative to /root/reference).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------
# functional building blocks
# --------------------------------------------------------------------------------------
def conv3x3(cin, cout, k=3):
    """rumpy/SISR/models/advanced/common.py:6-9 (default_conv): same-padding conv + bias."""
    return nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)


class ScaledResidualBlock(nn.Module):
    """rumpy/SISR/models/advanced/common.py:51-75 (ResBlock): x + res_scale * conv(relu(conv(x))).

    state_dict keys ``body.0.*`` / ``body.2.*`` (index 1 is the shared in-place ReLU).
    """

    def __init__(self, feats, res_scale):
        super().__init__()
        self.body = nn.Sequential(conv3x3(feats, feats), nn.ReLU(), conv3x3(feats, feats))
        self.res_scale = res_scale

    def forward(self, x):
        return self.body(x) * self.res_scale + x


def make_upsampler(scale, feats):
    """rumpy/SISR/models/advanced/common.py:23-48 (Upsampler, act=False, bn=False).

    scale = 2^n: n x [conv feats->4*feats, PixelShuffle(2)]; scale = 3: conv feats->9*feats, PixelShuffle(3).
    """
    layers = []
    if scale & (scale - 1) == 0:
        for _ in range(int(round(math.log2(scale)))):
            layers += [conv3x3(feats, 4 * feats), nn.PixelShuffle(2)]
    elif scale == 3:
        layers += [conv3x3(feats, 9 * feats), nn.PixelShuffle(3)]
    else:
        raise NotImplementedError
    return nn.Sequential(*layers)


class ChannelAttention(nn.Module):
    """rumpy/SISR/models/advanced/architectures.py:24-44 (CALayer).

    x * sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2); keys ``conv_du.0.*`` / ``conv_du.2.*``.
    """

    def __init__(self, feats, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(
            nn.Conv2d(feats, feats // reduction, 1), nn.ReLU(),
            nn.Conv2d(feats // reduction, feats, 1), nn.Sigmoid())

    def forward(self, x):
        return x * self.conv_du(self.avg_pool(x))


class AttentionResidualBlock(nn.Module):
    """rumpy/SISR/models/advanced/architectures.py:60-84 (RCAB).

    x + CA(conv(relu(conv(x)))).  NOTE: the reference stores ``res_scale`` but never applies it
    (architectures.py:79-84) - reproduced here.
    """

    def __init__(self, feats, reduction, res_scale=1):
        super().__init__()
        self.body = nn.Sequential(conv3x3(feats, feats), nn.ReLU(), conv3x3(feats, feats),
                                  ChannelAttention(feats, reduction))
        self.res_scale = res_scale  # unused, as in the reference

    def forward(self, x):
        return self.body(x) + x


class AttentionGroup(nn.Module):
    """rumpy/SISR/models/advanced/architectures.py:107-124 (ResidualGroup): n RCAB + conv, + skip."""

    def __init__(self, feats, reduction, res_scale, n_blocks):
        super().__init__()
        mods = [AttentionResidualBlock(feats, reduction, res_scale) for _ in range(n_blocks)]
        mods.append(conv3x3(feats, feats))
        self.body = nn.Sequential(*mods)

    def forward(self, x):
        return self.body(x) + x


class OracleEDSR(nn.Module):
    """rumpy/SISR/models/advanced/architectures.py:198-241 (EDSR).

    head conv -> num_blocks x ResBlock -> conv -> + head output -> Upsampler -> conv.
    Keys: head.0, body.{i}.body.{0,2}, body.{num_blocks}, tail.0.{0,2,..}, tail.1
    """

    def __init__(self, in_features=3, out_features=3, net_features=64, num_blocks=16, scale=4, res_scale=0.1):
        super().__init__()
        f = net_features
        self.head = nn.Sequential(conv3x3(in_features, f))
        self.body = nn.Sequential(*([ScaledResidualBlock(f, res_scale) for _ in range(num_blocks)]
                                    + [conv3x3(f, f)]))
        self.tail = nn.Sequential(make_upsampler(scale, f), conv3x3(f, out_features))

    def forward(self, x):
        x = self.head(x)
        return self.tail(self.body(x) + x)


class OracleRCAN(nn.Module):
    """rumpy/SISR/models/advanced/architectures.py:140-176 (RCAN).

    head conv -> n_resgroups x ResidualGroup -> conv -> + head output -> Upsampler -> conv.
    Keys: head.0, body.{g}.body.{b}.body.{0,2}, body.{g}.body.{b}.body.3.conv_du.{0,2},
    body.{g}.body.{n_resblocks}, body.{n_resgroups}, tail.0.{0,2}, tail.1
    """

    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4,
                 reduction=16, res_scale=1.0, **_ignored):
        super().__init__()
        f = n_feats
        self.head = nn.Sequential(conv3x3(in_feats, f))
        self.body = nn.Sequential(*([AttentionGroup(f, reduction, res_scale, n_resblocks)
                                     for _ in range(n_resgroups)] + [conv3x3(f, f)]))
        self.tail = nn.Sequential(make_upsampler(scale, f), conv3x3(f, out_feats))

    def forward(self, x):
        x = self.head(x)
        return self.tail(self.body(x) + x)


class MetaAttention(nn.Module):
    """rumpy/SISR/models/attention_manipulators/q_layer.py:5-45 (ParaCALayer, the meta-attention block), nonlinearity=True:
    x * sigmoid(FC_k(...relu(FC_1(metadata)))) with 1x1 convs on the [N,M,1,1] vector.  Unit sizes q_layer.py:27-31:
    layer i of num_layers maps to channels // (num_layers - i) (or (channels - M) // (num_layers - i) + M when M > 15).
    Keys ``attribute_integrator.{0,2,..}.*`` (ReLU after every layer but the last, then Sigmoid)."""

    def __init__(self, feats, num_metadata, num_layers=2):
        super().__init__()
        layers, sizes, mult = [], [num_metadata], num_layers
        for i in range(num_layers):
            sizes.append((feats - num_metadata) // mult + num_metadata if num_metadata > 15 else feats // mult)
            layers.append(nn.Conv2d(sizes[i], sizes[i + 1], 1))
            if mult != 1:
                layers.append(nn.ReLU())
            mult -= 1
        layers.append(nn.Sigmoid())
        self.attribute_integrator = nn.Sequential(*layers)

    def forward(self, x, attributes):
        return x * self.attribute_integrator(attributes)


class StyledChannelAttention(nn.Module):
    """rumpy/SISR/models/attention_manipulators/architectures.py:41-136 (QCALayer) for the styles whose squeeze-excite MLP also reads the
    attribute vector: 'max_concat' (:116-117), 'mini_concat' (:118-120), 'extended_attention' (:121-124), 'softmax' (:125-127).
    Modules, creation order and keys as :66-108; forward(x, attributes [N,M,1,1]) -> x * gate."""

    def __init__(self, feats, reduction, style, num_metadata):
        super().__init__()
        if reduction < 16:
            raise RuntimeError('Using an extreme channel attention reduction value')
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.style, cr = style, feats // reduction
        cin = feats if style == 'mini_concat' else feats + num_metadata
        if style in ('max_concat', 'softmax'):
            self.conv_du = nn.Sequential(nn.Conv2d(cin, cr, 1), nn.ReLU(), nn.Conv2d(cr, feats, 1), nn.Sigmoid())
        elif style == 'mini_concat':
            self.pre_concat = nn.Conv2d(cin, cr, 1)
            self.conv_du = nn.Sequential(nn.ReLU(), nn.Conv2d(cr + num_metadata, feats, 1), nn.Sigmoid())
        elif style == 'extended_attention':
            self.feature_convs = nn.ModuleList(nn.Sequential(nn.Conv2d(i, o, 1), nn.ReLU()) for i, o in
                                               ((cin, feats // 2), (feats // 2 + num_metadata, feats // 4), (feats // 4 + num_metadata, cr)))
            self.final_conv = nn.Sequential(nn.Conv2d(cr, feats, 1), nn.Sigmoid())
        else:
            raise NotImplementedError(style)
        if style == 'softmax':
            self.softmax = nn.Softmax(dim=1)

    def forward(self, x, attributes):
        y = self.avg_pool(x)
        if self.style == 'max_concat':
            y = self.conv_du(torch.cat((y, attributes), dim=1))
        elif self.style == 'mini_concat':
            y = self.conv_du(torch.cat((self.pre_concat(y), attributes), dim=1))
        elif self.style == 'extended_attention':
            for sec in self.feature_convs:
                y = sec(torch.cat((y, attributes), dim=1))
            y = self.final_conv(y)
        else:
            y = self.softmax(self.conv_du(torch.cat((y, attributes), dim=1)))
        return x * y


STYLED = ('max_concat', 'mini_concat', 'extended_attention', 'softmax')


class MetaAttentionResidualBlock(nn.Module):
    """rumpy/SISR/models/attention_manipulators/architectures.py:154-228 (QRCAB) for style='standard', q_layer=True, no other
    optional node: res = body(x); res = QCALayer_standard(res) (:113-136 -> plain channel attention, keys final_body.conv_du.*);
    res = q_node(res, metadata); res += x.  ``res_scale`` is stored and unused, as in the reference."""

    def __init__(self, feats, reduction, num_metadata, num_layers_in_q_layer=2, q_layer=True, style='standard'):
        super().__init__()
        self.style = style          # 'modulate' (QCALayer.forward :113-115): the attention vector is multiplied by the [N,C,1,1] attributes
        # creation order as the reference (architectures.py:166-172: the two convs first - same seed, same initial weights);
        # registration order as the reference (:173-195: final_body, q_node, body) - it is the order of the state_dict keys
        # and of the optimizer's parameter indices in a checkpoint
        convs = [conv3x3(feats, feats), nn.ReLU(), conv3x3(feats, feats)]
        self.final_body = StyledChannelAttention(feats, reduction, style, num_metadata) if style in STYLED else ChannelAttention(feats, reduction)
        self.q_layer = q_layer
        if q_layer:
            self.q_node = MetaAttention(feats, num_metadata, num_layers_in_q_layer)
        self.body = nn.Sequential(*convs)

    def forward(self, xm):
        x, meta = xm
        res = self.body(x)
        if self.style == 'modulate':
            res = res * (self.final_body.conv_du(self.final_body.avg_pool(res)) * meta)
        elif self.style in STYLED:
            res = self.final_body(res, meta)
        else:
            res = self.final_body(res)
        if self.q_layer:
            res = self.q_node(res, meta)
        return res + x, meta


class MetaAttentionGroup(nn.Module):
    """architectures.py:249-299 (QResidualGroup): n QRCABs (the first num_q_layers of them with a q-layer, all if None),
    then ``final_body`` conv, + skip."""

    def __init__(self, feats, reduction, n_blocks, num_metadata, q_layer, num_q_layers, num_layers_in_q_layer, style='standard'):
        super().__init__()
        blocks = [MetaAttentionResidualBlock(feats, reduction, num_metadata, num_layers_in_q_layer,
                                             q_layer=q_layer and (num_q_layers is None or b < num_q_layers), style=style) for b in range(n_blocks)]
        self.final_body = conv3x3(feats, feats)          # registered before body (architectures.py:292-293)
        self.body = nn.Sequential(*blocks)

    def forward(self, xm):
        res, _ = self.body(xm)
        return self.final_body(res) + xm[0], xm[1]


class OracleQRCAN(nn.Module):
    """architectures.py:313-462 (QRCAN) for style='standard' with the meta-attention q-layers (include_q_layer=True;
    selective_meta_blocks / num_q_layers_inner_residual honoured), no pixel attention / dgfmb / sft / da-conv nodes.
    forward(x, metadata [N,M,1,1]).  Keys: head.0, body.{g}.body.{b}.body.{0,2}, body.{g}.body.{b}.final_body.conv_du.{0,2},
    body.{g}.body.{b}.q_node.attribute_integrator.{0,2}, body.{g}.final_body, final_body, tail.0.{0,2}, tail.1"""

    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4, reduction=16,
                 num_metadata=1, include_q_layer=True, selective_meta_blocks=None, num_q_layers_inner_residual=None,
                 num_layers_in_q_layer=2, style='standard', **_ignored):
        super().__init__()
        if style not in ('standard', 'modulate') + STYLED:
            raise NotImplementedError('oracle: unknown QCALayer style %r' % style)
        f = n_feats
        head = conv3x3(in_feats, f)                      # creation order (random init) and registration order (keys) of
        groups = [MetaAttentionGroup(                    # architectures.py:368-433: head, groups, final_body, tail created;
            f, reduction, n_resblocks, num_metadata,     # final_body, head, body, tail registered
            include_q_layer and (selective_meta_blocks is None or bool(selective_meta_blocks[g])),
            num_q_layers_inner_residual, num_layers_in_q_layer, style=style) for g in range(n_resgroups)]
        self.final_body = conv3x3(f, f)
        tail = [make_upsampler(scale, f), conv3x3(f, out_feats)]
        self.head = nn.Sequential(head)
        self.body = nn.Sequential(*groups)
        self.tail = nn.Sequential(*tail)

    def forward(self, x, metadata):
        x = self.head(x)
        res, _ = self.body((x, metadata))
        return self.tail(self.final_body(res) + x)


class _Bf16Point(torch.autograd.Function):
    """a 16-bit storage point of the HIP pipeline inside an fp32 graph: the value is rounded on the way forward (to `fwd` = True: bf16, or a
    torch dtype), its gradient to bf16 on the way back (each optional)"""

    @staticmethod
    def forward(ctx, t, fwd, bwd):
        ctx.bwd = bwd
        if fwd is True:
            fwd = torch.bfloat16
        return t.to(fwd).float() if fwd else t

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).float() if ctx.bwd else g), None, None


class OracleEncoder(nn.Module):
    """rumpy/regression/models/contrastive_learning/encoding_models.py:5-55 (Encoder, dropdown_q=None): six 3x3 convs
    (3-64-64-128/2-128-256/2-256), each + BatchNorm2d + LeakyReLU(0.1), global average pool -> ``fea`` [N,256]; ``mlp`` 256-256-256
    -> ``q``.  Returns (fea, {'q': q}).  Keys ``E.{0,1,3,4,...,16}.*`` and ``mlp.{0,2}.*``.

    ``bf16_storage = True`` (the name is round 2's) evaluates the SAME graph with the HIP path's storage precision: filters of convs 1..5,
    every conv output and every stage output rounded to ``storage_dtype`` on the way forward - IEEE fp16 since round 3, bf16 before -, their
    gradients to bf16 on the way back (fp32 accumulation everywhere, like the MFMA path).  A diagnostic, not the acceptance criterion: the
    training tests assert against the fp32 graph first (a LeakyReLU input that changes sign under the storage rounding changes its gradient
    tenfold: bf16's 2^-9 turns into a 5-10 % perturbation of this network's gradient, fp16's 2^-12 into 2-3 %), and against this variant
    tightly (same arithmetic, other summation order)."""

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for cout, stride in ((64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1)):
            layers += [nn.Conv2d(cin, cout, 3, stride=stride, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU(0.1)]
            cin = cout
        layers.append(nn.AdaptiveAvgPool2d(1))
        self.E = nn.Sequential(*layers)
        self.mlp = nn.Sequential(nn.Linear(256, 256), nn.LeakyReLU(0.1), nn.Linear(256, 256))
        self.bf16_storage = False
        self.storage_dtype = torch.float16
        self.z_fp32 = True          # round 4: conv outputs fp32, filters as image + rounding-residual image (False: the round-3 all-fp16 pass)

    def _trunk_bf16_storage(self, x):
        a = x
        for i in range(6):
            conv, bn = self.E[3 * i], self.E[3 * i + 1]
            sd = self.storage_dtype
            if i == 0:
                w = conv.weight                                            # the first conv reads fp32 filters on the HIP path too
            elif self.z_fp32:
                # round 4: the filter enters as its fp16 image + the fp16 image of its rounding residual (22 significant bits) ...
                hi = conv.weight.detach().to(sd).float()
                w = conv.weight + ((hi + (conv.weight.detach() - hi).to(sd).float()) - conv.weight.detach())
            else:
                w = _Bf16Point.apply(conv.weight, sd, False)
            z = F.conv2d(a, w, conv.bias, stride=conv.stride, padding=1)
            # ... and the conv output stays fp32 (its gradient is still stored as bf16)
            z = _Bf16Point.apply(z, None if self.z_fp32 else sd, True)
            a = _Bf16Point.apply(F.leaky_relu(bn(z), 0.1), sd, i < 5)      # the pool's gradient reaches the last stage in fp32
        return a.mean((2, 3))

    def forward(self, x):
        fea = self._trunk_bf16_storage(x) if self.bf16_storage else self.E(x).squeeze(-1).squeeze(-1)
        return fea, {'q': self.mlp(fea)}


class OracleBlindPipeline(nn.Module):
    """rumpy/SISR/models/blur_kernel_blind_sr/contrastive_blind_sr.py:90-329 (ContrastiveBlindSRPipeline) for the default
    configuration: contrastive_encoder='default', embedding_type='pre-q' (embedding = E(x)[0], :244), encoder_freeze_mode='all'
    (:46-48: every encoder parameter has requires_grad False), no auxiliary encoder / reducer / normalisation / SFT:
    sr = G(x, embedding[:, :, None, None]) (:285,315).  Registration order G, E (:134,154).

    NOTE on BatchNorm mode, reproduced because it is what the reference computes: ContrastiveBlindQRCANHandler.run_train
    (blur_kernel_blind_sr/handlers.py:513-521) puts E in eval mode and then calls BaseModel.run_train, whose first statement is
    ``self.net.train()`` (base_architecture.py:472) - so during training the frozen encoder normalises with BATCH statistics and
    keeps updating its running statistics; run_eval (``self.net.eval()``, :503) uses those running statistics."""

    def __init__(self, generator, embedding_type='pre-q'):
        super().__init__()
        self.G = generator
        self.E = OracleEncoder()
        self.embedding_type = embedding_type          # :134-145: 'pre-q' = E(x)[0], 'q' = E(x)[1]['q']
        for p in self.E.parameters():
            p.requires_grad = False

    def forward(self, x):
        fea, out = self.E(x)
        emb = fea if self.embedding_type == 'pre-q' else out['q']
        return self.G(x, emb.unsqueeze(2).unsqueeze(3))


def seeded_encoder_state(encoder, seed):
    """Deterministic, well-conditioned encoder state in state_dict order: conv / linear weights and biases U(-b, b) with
    b = sqrt(6 / fan_in) (He gain, so the six LeakyReLU stages keep O(1) activations), BN weight and running_var U(0.5, 1.5), BN bias and
    running_mean U(-0.2, 0.2), num_batches_tracked 0."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    ref = encoder.state_dict()
    for k, v in ref.items():
        shape = tuple(v.shape)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif len(shape) >= 2 or (k.endswith('bias') and ref[k[:-4] + 'weight'].dim() >= 2):
            w = ref[k] if len(shape) >= 2 else ref[k[:-4] + 'weight']
            b = math.sqrt(6.0 / (w[0].numel()))
            sd[k] = torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))
        elif k.endswith('weight') or k.endswith('running_var'):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, size=shape).astype(np.float32))
        else:
            sd[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, size=shape).astype(np.float32))
    return sd


def seeded_pipeline_state(pipe, seed):
    """G from seeded_state_dict(seed), E from seeded_encoder_state(seed + 1); keys as the pipeline's state_dict."""
    sd = OrderedDict(('G.' + k, v) for k, v in seeded_state_dict(pipe.G, seed).items())
    sd.update(('E.' + k, v) for k, v in seeded_encoder_state(pipe.E, seed + 1).items())
    return sd


# --------------------------------------------------------------------------------------
# handler-level restatement: one train step / one eval step
# --------------------------------------------------------------------------------------
class OracleSRCNN(nn.Module):
    """rumpy/SISR/models/basic/architectures.py:6-60: conv_0 .. conv_{d-1} (nn.Conv2d, padding k//2 for 'same'), F.relu between them,
    none after the last; defaults 9-5-5 kernels over 1-64-32-1 channels.  Keys ``layer_dict.conv_<i>.weight|bias``."""
    residual = False

    def __init__(self, kernel_pattern=None, channel_pattern=None, padding='same'):
        super().__init__()
        kernel_pattern = [9, 5, 5] if kernel_pattern is None else list(kernel_pattern)
        channel_pattern = [1, 64, 32, 1] if channel_pattern is None else list(channel_pattern)
        pads = [k // 2 for k in kernel_pattern] if padding == 'same' else [0] * len(kernel_pattern)      # :26-29
        self.layer_dict = nn.ModuleDict()
        self.depth = len(kernel_pattern)
        for i, (k, pd) in enumerate(zip(kernel_pattern, pads)):
            self.layer_dict['conv_%d' % i] = nn.Conv2d(channel_pattern[i], channel_pattern[i + 1], kernel_size=k, padding=pd)

    def forward(self, x):
        out = x
        for i in range(self.depth):                       # :47-52
            out = self.layer_dict['conv_%d' % i](out)
            if i != self.depth - 1:
                out = F.relu(out)
        return out + x if self.residual else out          # VDSR :66-77: torch.add(out, residual)


class OracleVDSR(OracleSRCNN):
    residual = True


class OracleHandler:
    """Restates BaseModel for the L1 + Adam (+ optional per-batch scheduler) configuration.

    rumpy/shared_framework/models/base_architecture.py:
      :40        criterion = nn.L1Loss()
      :79-99     define_optimizer: Adam(lr, betas default (0.9, .999) / optimizer_params beta_1,beta_2)
      :101-117   define_scheduler 'cosine_annealing_warm_restarts' (T_0=restart_period, T_mult=t_mult,
                 eta_min=lr_min); also multi_step_lr / step_lr restated below
      :425-440   standard_update: zero_grad, backward, [clip_grad_norm_], optimizer.step, scheduler.step (PER BATCH)
      :457-485   run_train  -> (loss ndarray, out tensor)
      :488-520   run_eval   -> (out tensor, loss ndarray | None, None)
    """

    def __init__(self, net, lr=1e-4, scheduler=None, scheduler_params=None, optimizer_params=None,
                 grad_clip=None, eval_mode=False, criterion='l1'):
        self.net = net
        self.eval_mode = eval_mode
        self.grad_clip = None if grad_clip == 0 else grad_clip
        self.criterion = nn.MSELoss() if criterion == 'mse' else nn.L1Loss()      # basic/handlers.py:14,31 use nn.MSELoss
        self.optimizer = None
        self.learning_rate_scheduler = None
        if not eval_mode:
            params = [p for p in net.parameters() if p.requires_grad]
            if optimizer_params is not None:
                self.optimizer = torch.optim.Adam(params, lr=lr, betas=(optimizer_params['beta_1'],
                                                                         optimizer_params['beta_2']))
            else:
                self.optimizer = torch.optim.Adam(params, lr=lr)
            if scheduler == 'cosine_annealing_warm_restarts':
                self.learning_rate_scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(
                    self.optimizer, T_0=scheduler_params['restart_period'], T_mult=scheduler_params['t_mult'],
                    eta_min=scheduler_params['lr_min'])
            elif scheduler == 'multi_step_lr':
                self.learning_rate_scheduler = torch.optim.lr_scheduler.MultiStepLR(
                    self.optimizer, milestones=scheduler_params['milestones'], gamma=scheduler_params['gamma'])
            elif scheduler == 'step_lr':
                self.learning_rate_scheduler = torch.optim.lr_scheduler.StepLR(
                    self.optimizer, step_size=scheduler_params['step_size'], gamma=scheduler_params['gamma'])
            elif scheduler is not None:
                raise RuntimeError('%s scheduler not implemented' % scheduler)

    def run_train(self, x, y, scheduler_skip=False, extra_channels=None):
        """extra_channels: the [N,M,1,1] metadata tensor of the Q models (attention_manipulators/__init__.py:186-202)"""
        if self.eval_mode:
            raise RuntimeError('Model initialized in eval mode, training not possible.')
        self.net.train()
        out = self.net(x) if extra_channels is None else self.net(x, extra_channels)
        loss = self.criterion(out, y)
        self.optimizer.zero_grad()
        loss.backward()
        if self.grad_clip is not None:
            nn.utils.clip_grad_norm_(self.net.parameters(), self.grad_clip)
        self.optimizer.step()
        if self.learning_rate_scheduler is not None and not scheduler_skip:
            self.learning_rate_scheduler.step()
        return loss.detach().cpu().numpy(), out.detach().cpu()

    def run_eval(self, x, y=None, request_loss=False, extra_channels=None):
        self.net.eval()
        with torch.no_grad():
            out = self.net(x) if extra_channels is None else self.net(x, extra_channels)
            loss = self.criterion(out, y).detach().cpu().numpy() if (request_loss and y is not None) else None
        return out.detach().cpu(), loss, None

    def get_learning_rate(self):
        return self.optimizer.param_groups[0]['lr']


def build_oracle(name, **internal_params):
    """Handler kwargs -> architecture kwargs, as the reference handlers map them.

    EDSRHandler rumpy/SISR/models/advanced/handlers.py:13-25: scale, in_features, num_features->net_features,
    num_blocks, res_scale (defaults 4, 3, 64, 16, 0.1).
    RCANHandler :33-42: scale, in_features->in_feats, remaining kwargs forwarded to RCAN(**kwargs).
    """
    p = dict(internal_params)
    if name == 'edsr':
        return OracleEDSR(in_features=p.get('in_features', 3), net_features=p.get('num_features', 64),
                          num_blocks=p.get('num_blocks', 16), scale=p.get('scale', 4),
                          res_scale=p.get('res_scale', 0.1))
    if name == 'rcan':
        fwd = {k: p[k] for k in ('n_resblocks', 'n_resgroups', 'n_feats', 'out_feats', 'reduction', 'res_scale')
               if k in p}
        return OracleRCAN(scale=p.get('scale', 4), in_feats=p.get('in_features', 3), **fwd)
    if name == 'qrcan':
        # QRCANHandler attention_manipulators/handlers.py:27-45: scale, in_features->in_feats, n_feats, style, num_metadata from
        # the metadata list (QModel, attention_manipulators/__init__.py:23-58), remaining kwargs forwarded to QRCAN(**kwargs)
        fwd = {k: p[k] for k in ('n_resblocks', 'n_resgroups', 'n_feats', 'out_feats', 'reduction', 'include_q_layer',
                                 'selective_meta_blocks', 'num_q_layers_inner_residual', 'num_layers_in_q_layer') if k in p}
        return OracleQRCAN(scale=p.get('scale', 4), in_feats=p.get('in_features', 3), style=p.get('style', 'standard'),
                           num_metadata=p.get('num_metadata', 1), **fwd)
    if name == 'contrastiveblindqrcan':
        # ContrastiveBlindQRCANHandler blur_kernel_blind_sr/handlers.py:455-510: QRCAN(num_metadata=encoder_output_size=256, ...) inside
        # ContrastiveBlindSRPipeline
        q = dict(p)
        q['num_metadata'] = p.get('encoder_output_size', 256)
        return OracleBlindPipeline(build_oracle('qrcan', **q), embedding_type=p.get('embedding_type', 'pre-q'))
    if name in ('srcnn', 'vdsr'):
        # SRCNNHandler / VDSRHandler basic/handlers.py:6-35 (VDSR defaults: 20 x 3x3, 1-64..64-1 channels)
        kp, cp = p.get('kernel_pattern'), p.get('channel_pattern')
        if name == 'vdsr':
            kp = [3] * 20 if kp is None else kp
            cp = [1] + [64] * 19 + [1] if cp is None else cp
        return (OracleVDSR if name == 'vdsr' else OracleSRCNN)(kernel_pattern=kp, channel_pattern=cp, padding=p.get('padding', 'same'))
    raise KeyError(name)


def scale_qpi(qpi, n_feats=64, min_mu=-0.2, max_mu=0.8, clamp=False, sig=0.2):
    """QRCANHandler.scale_qpi / gaussian (attention_manipulators/handlers.py:59-73), style 'modulate': a scalar q in [0,1] per image
    -> a gaussian bump over the channel axis, centre mu = q * (max_mu - min_mu) + min_mu on linspace(0, 1, n_feats), sigma 0.2,
    evaluated in float64 numpy and cast to float32.  qpi [N,1,1,1] -> [N,n_feats,1,1]."""
    base = np.linspace(0, 1, n_feats)
    scaled = (qpi * (max_mu - min_mu)) + min_mu
    rows = []
    for i in range(scaled.size(0)):
        mu = scaled[i].squeeze().numpy()
        rows.append(torch.from_numpy((1 / (np.sqrt(2 * np.pi) * sig)) * np.exp(-np.power(base - mu, 2.) / (2 * np.power(sig, 2.)))).type(torch.float32))
    full = torch.stack(rows)
    if clamp:
        full = torch.clamp(full, 0, 1)
    return full.unsqueeze(2).unsqueeze(3)


# --------------------------------------------------------------------------------------
# eval post-processing and metric (defines "eval PSNR")
# --------------------------------------------------------------------------------------
def clip01(im):
    """rumpy/shared_framework/models/base_interface.py:216-222 (_standard_image_formatting)."""
    return np.clip(np.copy(im), 0, 1)


def rgb_to_ycbcr_jpg(img, max_val=1):
    """rumpy/image_tools/image_manipulation/image_functions.py:81-96 ('jpg' matrix, y_only=False).

    img: C,H,W array.  Returns 3,H,W [Y, Cb, Cr].
    """
    bias_c = 128. * (max_val / 255)
    y = 0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]
    cb = bias_c + (-0.168736 * img[0] - 0.331264 * img[1] + 0.5 * img[2])
    cr = bias_c + (0.5 * img[0] - 0.418688 * img[1] - 0.081312 * img[2])
    return np.array([y, cb, cr])


def net_run_and_process(handler, lr, hr=None, request_loss=False):
    """rumpy/SISR/models/interface.py:103-124, 'rgb' colourspace branch.

    -> (rgb clipped [B,3,H,W] ndarray, ycbcr [B,3,H,W] ndarray of the clipped rgb, loss, timing)
    """
    out_rgb, loss, timing = handler.run_eval(lr, hr, request_loss=request_loss)
    clipped = clip01(out_rgb.numpy())
    ycbcr = np.copy(clipped)
    for i in range(ycbcr.shape[0]):
        ycbcr[i] = rgb_to_ycbcr_jpg(ycbcr[i])
    return clipped, ycbcr, loss, timing


def psnr(img1, img2, max_value=1.0):
    """rumpy/sr_tools/metrics.py:33-44: float32 mse, 100 if identical, 20 log10(max/sqrt(mse))."""
    mse = np.mean((np.array(img1, dtype=np.float32) - np.array(img2, dtype=np.float32)) ** 2)
    if mse == 0:
        return 100
    return 20 * np.log10(max_value / (np.sqrt(mse)))


def y_psnr(ycbcr_a, ycbcr_ref):
    """rumpy/sr_tools/metrics.py:109-121 (run_psnr, multichannel=False): PSNR over channel 0 of the whole batch,
    max_value=1, no border shave."""
    return psnr(ycbcr_a[:, 0, :, :], ycbcr_ref[:, 0, :, :], max_value=1)


# --------------------------------------------------------------------------------------
# deterministic synthetic tensors (regenerable on the GPU box without the reference)
# --------------------------------------------------------------------------------------
def seeded_state_dict(module, seed, scale=None):
    """Fill every parameter from numpy.random.default_rng(seed) in state_dict order.

    Weights ~ U(-b, b) with b = 1/sqrt(fan_in) (the magnitude of torch's default conv init, so deep stacks
    stay well-conditioned); biases likewise.  ``scale`` optionally overrides b.
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, v in module.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith('weight') and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
        elif k.endswith('bias'):
            w = module.state_dict()[k[:-4] + 'weight']
            fan_in = w.shape[1] * w.shape[2] * w.shape[3]
        else:
            fan_in = 1
        b = scale if scale is not None else 1.0 / math.sqrt(fan_in)
        sd[k] = torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))
    return sd


def synthetic_batch(seed, n, lr_hw=48, scale=4, channels=3):
    """x uniform[0,1) fp32 [n,C,h,w], y uniform[0,1) fp32 [n,C,s*h,s*w], default_rng(seed)."""
    rng = np.random.default_rng(seed)
    h, w = (lr_hw, lr_hw) if isinstance(lr_hw, int) else lr_hw
    x = torch.from_numpy(rng.random((n, channels, h, w), dtype=np.float32))
    y = torch.from_numpy(rng.random((n, channels, h * scale, w * scale), dtype=np.float32))
    return x, y


def bf16_round(t):
    """Round-to-nearest-even to bf16 and back to fp32 (what the MFMA operands see)."""
    return t.to(torch.bfloat16).to(torch.float32)


_QUAD_TAPS = ((0.15625, 0.9375, -0.09375), (-0.09375, 0.9375, 0.15625))


def interpolating_state_dict(module, seed, noise=0.03, detail=0.02, body_gain=1.0):
    """Seeded weights that put an EDSR / RCAN (x4, 64 features) into the regime of a TRAINED super-resolver (Y-PSNR >= 30 dB on
    natural images) without shipping trained weights: the network computes a two-stage quadratic x4 interpolation of its input
    plus a small detail term to which every body convolution contributes.  Test data, not a restatement of reference code
    (the reference's module classes are filled with it through load_state_dict; fixtures G17 / G18).

      head.0    : channels >= 3: U(-b, b) features (b = 1/sqrt(fan_in), as seeded_state_dict); channel c < 3: centre tap 1 from colour c
                  + U(-b, b) * noise
      body.*    : U(-b, b) * body_gain - every RCAB / ResBlock / group conv at the magnitude of the default initialisation
      body[-1]  : U(-b, b) * detail - the body's output reaches the trunk as a small "detail" term next to the global skip
      tail.0.{0,2}: a separable x2 interpolation kernel (3-point Lagrange taps at -0.25 / +0.25) from channel c to sub-pixel channel 4c + 2i + j
                  for every c, + U(-b, b) * noise (no weight is exactly representable in bf16)
      tail.1    : centre tap 1 from channel c to colour c, + U(-b, b) * noise
    Everything comes from numpy.random.default_rng(seed) in state_dict order."""
    base = seeded_state_dict(module, seed)
    keys = list(base.keys())
    body_ids = sorted({int(k.split('.')[1]) for k in keys if k.startswith('body.')})
    last_body = 'body.%d.' % body_ids[-1]
    sd = OrderedDict()
    for k, v in base.items():
        v = v.clone()
        if k.startswith('head.0.'):
            v[:3] *= noise              # the three colour-carrying channels: identity + a little of everything
            if k.endswith('weight'):
                for c in range(min(3, v.shape[1])):
                    v[c, c, 1, 1] += 1.0
        elif k.startswith(last_body):
            v *= detail
        elif k.startswith('body.'):
            v *= body_gain
        elif k.startswith('tail.0.'):
            v *= noise
            if k.endswith('weight'):
                f = v.shape[1]
                for c in range(f):
                    for i in range(2):
                        for j in range(2):
                            # sub-pixel row 2h+i sits at h - 0.25 (i = 0) / h + 0.25 (i = 1): quadratic (3-point Lagrange) taps on rows h-1, h, h+1
                            for ky in range(3):
                                for kx in range(3):
                                    v[4 * c + 2 * i + j, c, ky, kx] += _QUAD_TAPS[i][ky] * _QUAD_TAPS[j][kx]
        elif k.startswith('tail.1.'):
            v *= noise
            if k.endswith('weight'):
                for c in range(v.shape[0]):
                    v[c, c, 1, 1] += 1.0
        sd[k] = v
    return sd


def vignetted_pair(hr_u8, lr_hw, scale=4, margin=24):
    """Evaluation pair for the G17 / G18 fixtures from an RGB uint8 image: a centred (scale*lr_hw)^2 crop multiplied by a raised-cosine
    vignette (zero at the border over `margin` pixels, so that the zero padding of the convolutions is exact there), quantised to
    uint8 = HR; LR = PIL bicubic downsample of it (the reference's own resize, image_functions.py:13-41).  Returns (lr_u8, hr_u8)."""
    from PIL import Image
    n = scale * lr_hw
    y0, x0 = (hr_u8.shape[0] - n) // 2, (hr_u8.shape[1] - n) // 2
    crop = hr_u8[y0:y0 + n, x0:x0 + n].astype(np.float64) / 255.0
    w = np.ones(n)
    ramp = 0.5 - 0.5 * np.cos(np.pi * (np.arange(margin) + 0.5) / margin)
    w[:margin], w[-margin:] = ramp, ramp[::-1]
    hr = (np.clip(crop * w[:, None, None] * w[None, :, None], 0, 1) * 255 + 0.5).astype(np.uint8)
    lr = np.asarray(Image.fromarray(hr).resize((lr_hw, lr_hw), Image.BICUBIC))
    return lr, hr


def forward_chop(run, x, scale, max_combined_im_size, shave=10):
    """Tiled whole-image evaluation as the reference has it twice: SANHandler.forward_chop (rumpy/SISR/models/advanced/handlers.py:85-123) and
    ContrastiveBlindQEDSRHandler.forward_chop (rumpy/SISR/models/blur_kernel_blind_sr/handlers.py:907-945).  Restated from what that code does:
    the image is covered by four corner-anchored windows of (H // 2 + shave) x (W // 2 + shave) pixels; each is super-resolved by
    `run(chunk)` (the handlers' run_eval(...)[0]) - or, while a window still has max_combined_im_size pixels or more, by this function again -
    and contributes the output pixels of its own quadrant (rows / columns below or from H // 2 / W // 2).  Pinned on the reference's own
    method by fixture G23 (tests/golden/make_golden_chop.py)."""
    n, c, height, width = x.shape
    win_h, win_w = height // 2 + shave, width // 2 + shave
    deeper = win_h * win_w >= max_combined_im_size
    out = x.new_empty(n, c, scale * height, scale * width)
    for top in (True, False):
        y0 = 0 if top else height - win_h                       # window rows [y0, y0 + win_h)
        ys = (0, height // 2) if top else (height // 2, height)   # output rows this window owns (LR units)
        for left in (True, False):
            x0 = 0 if left else width - win_w
            xs = (0, width // 2) if left else (width // 2, width)
            chunk = x[:, :, y0:y0 + win_h, x0:x0 + win_w]
            sr = forward_chop(run, chunk, scale, max_combined_im_size, shave=shave) if deeper else run(chunk)
            out[:, :, scale * ys[0]:scale * ys[1], scale * xs[0]:scale * xs[1]] = \
                sr[:, :, scale * (ys[0] - y0):scale * (ys[1] - y0), scale * (xs[0] - x0):scale * (xs[1] - x0)]
    return out
