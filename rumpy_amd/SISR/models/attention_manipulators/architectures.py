"""QRCAN with meta-attention for the MI355X HIP path - drop-in for rumpy/SISR/models/attention_manipulators/architectures.py
(QRCAN :313-462, QResidualGroup :249-299, QRCAB :154-228, QCALayer :41-136) and q_layer.py (ParaCALayer :5-45).

Supported configurations: ``style='standard'`` (plain channel attention in QCALayer) with ``include_q_layer=True`` -
the meta-attention setup of the reference's blind / non-blind QRCAN experiments: a ParaCALayer with two FC layers and ReLU after every
residual block (or the subsets selected by ``selective_meta_blocks`` / ``num_q_layers_inner_residual``) - and ``style='modulate'``, the
handler's default: the attention vector of every block is multiplied by a [N, 64] attribute vector (QCALayer.forward :113-115).
The other QCALayer styles ('max_concat', 'mini_concat', 'extended_attention', 'softmax': the gate MLP also reads the attribute vector) run as
separate launches around the block kernel (csrc/qca_style.hip; slower than the one-launch RCAB kernels, bf16 evaluation plans); the
pixel-attention / dgfmb / SFT / DA-conv nodes raise.

As in rumpy_amd/SISR/models/advanced/architectures.py the module tree only owns the parameters, under the reference's
state_dict keys AND in the reference's registration order (final_body before head/body/tail, a block's final_body and q_node
before its body: it fixes the order of the keys and of the optimizer's parameter indices in a checkpoint) and it creates the
layers in the reference's creation order (same seed -> same initial weights).  All arithmetic runs in the HIP engine:
gate = channel attention x meta attention inside the fused channel-attention kernels, the metadata MLPs of all blocks in one
launch (rumpy_q_mlp_fwd / rumpy_q_mlp_bwd_params).
"""
from torch import nn

from rumpy_amd.engine import NetSpec, StyledCAParams
from rumpy_amd.SISR.models.advanced.architectures import HipSRNet, _CAParams, _conv, _upsampler


class _QLayerParams(nn.Module):
    # q_layer.py:22-45 with nonlinearity=True (as QRCAB builds it, architectures.py:182-183): attribute_integrator = [conv1x1, ReLU] x (num_layers - 1),
    # conv1x1, Sigmoid; layer i maps to feats // (num_layers - i) units, or (feats - M) // (num_layers - i) + M when M > 15
    def __init__(self, feats, num_metadata, num_layers=2):
        super().__init__()
        if not 1 <= num_layers <= 8:
            raise RuntimeError('rumpy_amd: q-layers with %d FC layers are not implemented on the HIP path (1 .. 8)' % num_layers)
        layers, sizes, mult = [], [num_metadata], num_layers
        for i in range(num_layers):
            sizes.append((feats - num_metadata) // mult + num_metadata if num_metadata > 15 else feats // mult)
            layers.append(nn.Conv2d(sizes[i], sizes[i + 1], 1, padding=0, bias=True))
            if mult != 1:
                layers.append(nn.ReLU(inplace=True))
            mult -= 1
        layers.append(nn.Sigmoid())
        self.attribute_integrator = nn.Sequential(*layers)
        self.num_layers = num_layers


STYLED = ('max_concat', 'mini_concat', 'extended_attention', 'softmax')      # QCALayer styles whose gate MLP also reads the attribute vector


class _QCAParams(nn.Module):
    # QCALayer.__init__ (architectures.py:47-110) for the styles in STYLED: same modules, same creation / registration order, same keys
    def __init__(self, feats, reduction, style, num_metadata):
        super().__init__()
        if reduction < 16:
            raise RuntimeError('Using an extreme channel attention reduction value')
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.style, cr = style, feats // reduction
        cin = feats if style == 'mini_concat' else feats + num_metadata
        if style in ('max_concat', 'softmax'):
            self.conv_du = nn.Sequential(nn.Conv2d(cin, cr, 1, padding=0, bias=True), nn.ReLU(inplace=True),
                                         nn.Conv2d(cr, feats, 1, padding=0, bias=True), nn.Sigmoid())
        elif style == 'mini_concat':
            self.pre_concat = nn.Conv2d(cin, cr, 1, padding=0, bias=True)
            self.conv_du = nn.Sequential(nn.ReLU(inplace=True), nn.Conv2d(cr + num_metadata, feats, 1, padding=0, bias=True), nn.Sigmoid())
        else:   # extended_attention
            self.feature_convs = nn.ModuleList(
                nn.Sequential(nn.Conv2d(i, o, 1, padding=0, bias=True), nn.ReLU(inplace=True))
                for i, o in ((cin, feats // 2), (feats // 2 + num_metadata, feats // 4), (feats // 4 + num_metadata, cr)))
            self.final_conv = nn.Sequential(nn.Conv2d(cr, feats, 1, padding=0, bias=True), nn.Sigmoid())
        if style == 'softmax':
            self.softmax = nn.Softmax(dim=1)

    def gate_layers(self):
        """[(conv module, cat, relu_in, act)] in evaluation order (QCALayer.forward :111-133); act: 0 none, 1 ReLU, 2 sigmoid, 3 sigmoid + softmax"""
        if self.style == 'max_concat':
            return [(self.conv_du[0], 1, 0, 1), (self.conv_du[2], 0, 0, 2)]
        if self.style == 'softmax':
            return [(self.conv_du[0], 1, 0, 1), (self.conv_du[2], 0, 0, 3)]
        if self.style == 'mini_concat':
            return [(self.pre_concat, 0, 0, 0), (self.conv_du[1], 1, 1, 2)]
        return [(m[0], 1, 0, 1) for m in self.feature_convs] + [(self.final_conv[0], 0, 0, 2)]


class _QRCABParams(nn.Module):
    # architectures.py:159-196: the two convs are created first, then QCALayer, then the q-node; registered: final_body, q_node, body
    def __init__(self, feats, reduction, num_metadata, q_layer, num_layers_in_q_layer, style='standard'):
        super().__init__()
        convs = [_conv(feats, feats), nn.ReLU(True), _conv(feats, feats)]
        self.final_body = _QCAParams(feats, reduction, style, num_metadata) if style in STYLED else _CAParams(feats, reduction)
        self.q_layer = q_layer
        if q_layer:
            self.q_node = _QLayerParams(feats, num_metadata, num_layers_in_q_layer)
        self.body = nn.Sequential(*convs)


class _QGroupParams(nn.Module):
    # architectures.py:254-294: blocks created, then the group conv; registered: final_body, body
    def __init__(self, feats, reduction, n_resblocks, num_metadata, q_layer, num_q_layers, num_layers_in_q_layer, style='standard'):
        super().__init__()
        blocks = [_QRCABParams(feats, reduction, num_metadata, q_layer and (num_q_layers is None or b < num_q_layers), num_layers_in_q_layer, style)
                  for b in range(n_resblocks)]
        self.final_body = _conv(feats, feats)
        self.body = nn.Sequential(*blocks)


class QRCAN(HipSRNet):
    def __init__(self, n_resblocks=20, n_resgroups=10, n_feats=64, in_feats=3, out_feats=3, scale=4, reduction=16, res_scale=1.0,
                 style='modulate', num_metadata=1, include_pixel_attention=False, selective_meta_blocks=None, include_q_layer=False,
                 num_q_layers_inner_residual=None, num_layers_in_q_layer=2, include_sft_layer=False, include_dgfmb_layer=False,
                 use_dgfmb_outer_reduction=False, include_da_conv_layer=False, staggered_encoding=False, **kwargs):
        super().__init__()
        unsupported = [k for k, v in (('style=%r' % style, style not in ('standard', 'modulate') + STYLED),
                                      ('style %r with more than 256 metadata entries' % style, style in STYLED and num_metadata > 256),
                                      ('style "modulate" together with q-layers', style == 'modulate' and include_q_layer),
                                      ('style "modulate" with n_feats != 64', style == 'modulate' and n_feats != 64),
                                      ('include_pixel_attention', include_pixel_attention),
                                      ('include_sft_layer', include_sft_layer), ('include_dgfmb_layer', include_dgfmb_layer),
                                      ('use_dgfmb_outer_reduction', use_dgfmb_outer_reduction), ('include_da_conv_layer', include_da_conv_layer),
                                      ('staggered_encoding', staggered_encoding)) if v]
        if unsupported:
            raise RuntimeError('rumpy_amd: QRCAN option(s) %s are not implemented on the HIP path (every QCALayer style, with or without '
                               'q-layers - "modulate" without); there is no fallback' % ', '.join(unsupported))
        f = n_feats
        self.supports_fused_l1 = f == 64          # (wider nets: the handlers take the generic loss path)
        self.scale, self.num_metadata, self.style = scale, num_metadata, style
        self.metadata_reduction = nn.Sequential(nn.Identity())
        head = _conv(in_feats, f)
        groups = [_QGroupParams(f, reduction, n_resblocks, num_metadata,
                                include_q_layer and (selective_meta_blocks is None or bool(selective_meta_blocks[g])),
                                num_q_layers_inner_residual, num_layers_in_q_layer, style) for g in range(n_resgroups)]
        self.final_body = _conv(f, f)
        tail = [_upsampler(scale, f), _conv(f, out_feats)]
        self.head = nn.Sequential(head)
        self.body = nn.Sequential(*groups)
        self.tail = nn.Sequential(*tail)
        self._finalize()

    def _styled_ca(self, name, ca):
        idx = {id(p): i for i, p in enumerate(self.param_list)}
        layers, n_prev = [], ca.gate_layers()[0][0].weight.shape[1] - (self.num_metadata if ca.gate_layers()[0][1] else 0)
        for conv, cat, relu_in, act in ca.gate_layers():
            layers.append(dict(w=conv.weight.data.reshape(conv.weight.shape[0], -1), b=conv.bias.data, gw=self.grad_views[idx[id(conv.weight)]],
                               gb=self.grad_views[idx[id(conv.bias)]], n_prev=n_prev, cat=cat, relu_in=relu_in, act=act))
            n_prev = conv.weight.shape[0]
        return StyledCAParams(name, layers, self.num_metadata)

    def _spec(self):
        if self.scale not in (1, 2, 3, 4, 8):
            raise RuntimeError('rumpy_amd: scale %s not supported by the HIP path' % self.scale)
        body, any_q = [], False
        for gi, grp in enumerate(self.body):
            items = []
            for bi, rb in enumerate(grp.body):
                pre = 'body.%d.body.%d' % (gi, bi)
                q = self._q_layer(pre + '.q_node', rb.q_node) if rb.q_layer else None
                any_q = any_q or q is not None
                ca = self._styled_ca(pre + '.final_body', rb.final_body) if self.style in STYLED else self._ca_layer(pre + '.final_body', rb.final_body)
                items.append(('rcab', self._conv_layer(pre + '.body.0', rb.body[0]), self._conv_layer(pre + '.body.2', rb.body[2]), ca, q))
            body.append(('group', items, self._conv_layer('body.%d.final_body' % gi, grp.final_body)))
        ups = [self._conv_layer('tail.0.%d' % i, m, shuffle=True) for i, m in enumerate(self.tail[0]) if isinstance(m, nn.Conv2d)]
        return NetSpec(self._conv_layer('head.0', self.head[0], kind='head'), body, self._conv_layer('final_body', self.final_body), ups,
                       self._conv_layer('tail.1', self.tail[1], kind='tail'), self.scale,
                       num_metadata=64 if self.style == 'modulate' else (self.num_metadata if (any_q or self.style in STYLED) else 0),
                       modulate=self.style == 'modulate')
