"""Golden fixture G20 (contrastive training of the degradation encoder: MoCo and SupMoCo steps) from the REAL reference
handlers.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_contrastive.py
define_model('mococontrastive') and define_model('supmoco') - the handlers of the reference's own contrastive tests
(automated_testing/contrastive_tests/test_contrastive_cpu_execute.py:33-50) - are built on the CPU with the default (DASR) encoder, their
encoders set from oracle.sr_oracle.seeded_encoder_state and their queues from oracle.contrastive_oracle.seeded_queue, and driven through
the reference's run_train for two steps each on oracle.contrastive_oracle.contrastive_batch inputs.  stored: losses, logits, every
gradient's norm and a strided sample of it, weights / running statistics / queue after the steps, and the class labels the reference's
class_logic assigns to a table of metadata rows under its three labelling strategies."""
import os
import runpy
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
shim = runpy.run_path(os.path.join(HERE, 'make_golden.py'), run_name='shim_only')
O = shim['O']
from oracle import contrastive_oracle as CO  # noqa: E402
from rumpy.shared_framework.models import define_model  # noqa: E402
from rumpy.regression.models.contrastive_learning import class_retrieval, partition_metadata, register_metadata, vector_retrieval  # noqa: E402
from rumpy.sr_tools.loss_functions import SupConLoss  # noqa: E402

STRIDE = 97
NOISE_KEYS = [('gaussian_noise_scale',), ('poisson_noise_scale',), ('gray_noise_boolean',)]
ALL_KEYS = NOISE_KEYS + [('jpeg_quality_factor',), ('jm_qpi',), ('realesrganblur-kernel_type',), ('realesrganblur-sigma_x',), ('realesrganblur-sigma_y',)]


def seed_net(net, seed):
    enc = O.seeded_encoder_state(O.OracleEncoder(), seed)
    net.encoder_q.load_state_dict(enc)
    net.encoder_k.load_state_dict(enc)
    net.queue = CO.seeded_queue(256, net.K, seed + 1)
    net.queue_ptr[0] = 0


def dump(d, tag, h):
    for k, p in h.net.named_parameters():
        if p.requires_grad and p.grad is not None:
            g = p.grad.detach().numpy().reshape(-1)
            d['%s.gnorm.%s' % (tag, k)] = np.asarray(np.linalg.norm(g.astype(np.float64)))
            d['%s.gsample.%s' % (tag, k)] = g[::STRIDE].copy()


def dump_state(d, tag, h, ncols):
    for k, v in h.net.state_dict().items():
        a = v.detach().numpy()
        if k.startswith('queue'):
            continue
        d['%s.sum.%s' % (tag, k)] = np.asarray(a.astype(np.float64).sum())
        d['%s.sample.%s' % (tag, k)] = a.reshape(-1)[::STRIDE].copy()
    d[tag + '.queue_head'] = h.net.queue[:, :ncols].numpy().copy()
    d[tag + '.queue_ptr'] = h.net.queue_ptr.numpy().copy()


def metadata_rows(seed, n):
    """rows over ALL_KEYS: [gaussian, poisson, gray, jpeg, jm, kernel type, sigma x, sigma y]"""
    rng = np.random.default_rng(seed)
    rows = np.zeros((n, 8), dtype=np.float32)
    for r in rows:
        if rng.random() < 0.5:
            r[0] = rng.uniform(0.02, 1)
        else:
            r[1] = rng.uniform(0.02, 1)
        r[2] = float(rng.random() < 0.5)
        if rng.random() < 0.5:
            r[3] = rng.uniform(0.02, 1)
        else:
            r[4] = rng.uniform(0.02, 1)
        r[5] = float(rng.integers(0, 7))
        r[6], r[7] = rng.uniform(0, 1, 2)
    return rows


def main():
    torch.manual_seed(0)
    d = {}
    # ---- MoCo, two crops per image (query | key on the channel axis) ----
    h = define_model('mococontrastive', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, model_name='default',
                     crop_count=2, lr=1e-3)
    seed_net(h.net, 2000)
    d['moco.keys'] = np.array(list(h.net.state_dict().keys()))
    for step in range(2):
        x = CO.contrastive_batch(2010 + step, 8, 2).view(8, 6, 32, 32)
        loss, logits = h.run_train(x=x, y=None)
        d['moco.loss%d' % step] = np.asarray(loss)
        d['moco.logits%d' % step] = logits.numpy()[:, :48].copy()
        d['moco.logits_rowsum%d' % step] = logits.numpy().astype(np.float64).sum(1)
        if step == 0:
            dump(d, 'moco.step0', h)
    dump_state(d, 'moco.after2', h, 16)
    # ---- MoCo, three crops (two positives per query) ----
    h = define_model('mococontrastive', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, model_name='default',
                     crop_count=3, lr=1e-3)
    seed_net(h.net, 2100)
    loss, logits = h.run_train(x=CO.contrastive_batch(2110, 4, 3).view(4, 9, 32, 32), y=None)     # crops on the channel axis
    d['moco3.loss0'] = np.asarray(loss)
    d['moco3.logits0'] = logits.numpy()[:, :48].copy()
    dump(d, 'moco3.step0', h)
    dump_state(d, 'moco3.after1', h, 4)
    # ---- SupMoCo, three crops, noise labels ----
    h = define_model('supmoco', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, model_name='default',
                     crop_count=3, lr=1e-3, data_type='noise', labelling_strategy='double_precision')
    seed_net(h.net, 2200)
    meta = np.array([[0.8, 0, 1], [0, 0.3, 0], [0.7, 0, 1], [0, 0.9, 1]], dtype=np.float32)      # images 0 and 2 share a class
    d['sup.meta'] = meta
    for step in range(2):
        x = CO.contrastive_batch(2210 + step, 4, 3).view(4, 9, 32, 32)
        loss, emb = h.run_train(x=x, y=torch.from_numpy(meta), metadata_keys=NOISE_KEYS)
        d['sup.loss%d' % step] = np.asarray(loss)
        d['sup.embedding%d' % step] = emb.numpy().copy()
        if step == 1:
            dump(d, 'sup.step1', h)
    d['sup.total_classes'] = np.asarray(h.total_classes)
    d['sup.queue_labels_head'] = h.net.queue_labels[:8].numpy().copy()
    dump_state(d, 'sup.after2', h, 8)
    # ---- class labels: the reference's decision logic on a table of rows, three strategies x two data selections ----
    rows = metadata_rows(2300, 64)
    d['labels.rows'] = rows
    for strategy in ('default', 'double_precision', 'triple_precision'):
        for tag, keys, sel in (('noise', NOISE_KEYS, 'noise'), ('all', ALL_KEYS, 'all')):
            names = register_metadata([k[0] for k in keys])
            m_map = {k: names.index(k) for k in names}
            fam, mags, total = partition_metadata(m_map, sel, labelling_strategy=strategy)
            lab = [class_retrieval(torch.from_numpy(r[:len(keys)]), fam, m_map, mags, total, labelling_strategy=strategy) for r in rows]
            d['labels.%s.%s' % (strategy, tag)] = np.asarray(lab, dtype=np.int64)
            d['labels.%s.%s.total' % (strategy, tag)] = np.asarray(int(total))
            d['vectors.%s.%s' % (strategy, tag)] = np.stack([vector_retrieval(torch.from_numpy(r[:len(keys)]), fam, m_map).numpy() for r in rows])
    # ---- WeakCon, three crops, noise vectors ----
    h = define_model('weakcon', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, model_name='default',
                     crop_count=3, lr=1e-3, data_type='noise')
    seed_net(h.net, 2400)
    for step in range(2):
        x = CO.contrastive_batch(2410 + step, 4, 3).view(4, 9, 32, 32)
        loss, emb = h.run_train(x=x, y=torch.from_numpy(meta), metadata_keys=NOISE_KEYS)
        d['weak.loss%d' % step] = np.asarray(loss)
        d['weak.embedding%d' % step] = emb.numpy().copy()
        if step == 1:
            dump(d, 'weak.step1', h)
    d['weak.queue_vectors_head'] = h.net.queue_vectors[:, :8].numpy().copy()
    dump_state(d, 'weak.after2', h, 8)
    # ---- SupConLoss on seeded features (the SupConHandler itself raises in the reference: handlers.py:248 indexes the encoder's output
    # dict as a tensor), value and gradient ----
    rng = np.random.default_rng(2500)
    feats = torch.from_numpy(rng.standard_normal((6, 3, 32)).astype(np.float32) * 0.3).requires_grad_(True)
    labels = torch.tensor([[0., 1., 0., 2., 1., 0.]])
    loss = SupConLoss()(feats, labels)
    loss.backward()
    d['supcon.features'], d['supcon.labels'] = feats.detach().numpy().copy(), labels.numpy()
    d['supcon.loss'], d['supcon.grad'] = np.asarray(loss.detach()), feats.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g20_contrastive_train.npz'), **d)
    print('wrote g20: moco losses', d['moco.loss0'], d['moco.loss1'], 'moco3', d['moco3.loss0'], 'supmoco', d['sup.loss0'], d['sup.loss1'],
          'classes', d['sup.total_classes'], 'queue labels', d['sup.queue_labels_head'])


if __name__ == '__main__' and 'joint' not in sys.argv[1:]:
    main()


# ---- G21: the blind-SR handler's joint SR + contrastive losses (combined_loss_mode 'moco' / 'supmoco') ----
JOINT_KW = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style='standard', include_q_layer=True,
                selective_meta_blocks=[True, False], num_q_layers_inner_residual=1)


def main_joint():
    torch.manual_seed(0)
    d = {}
    tmp = tempfile.mkdtemp()
    # a SupMoCo checkpoint of the reference, for the 'supmoco' joint mode (its queue labels only exist in a trained / registered queue)
    hs = define_model('supmoco', model_save_dir=tmp, device=torch.device('cpu'), eval_mode=False, model_name='default', crop_count=3, lr=1e-3,
                      data_type='noise', labelling_strategy='double_precision')
    seed_net(hs.net, 2600)
    meta = np.array([[0.8, 0, 1], [0, 0.3, 0], [0.7, 0, 1], [0, 0.9, 1]], dtype=np.float32)
    hs.run_train(x=CO.contrastive_batch(2610, 4, 3, hw=16).view(4, 9, 16, 16), y=torch.from_numpy(meta), metadata_keys=NOISE_KEYS)
    hs.save_model('enc')                 # (tests rebuild this checkpoint with the oracle's SupMoCo step from the same seeds)
    ckpt = os.path.join(tmp, 'enc_0')
    for tag, mode, crops, freeze, extra in (('moco_all', 'moco', 2, 'all', dict(block_encoder_loading=True)),
                                            ('moco_preq', 'moco', 2, 'pre_q', dict(block_encoder_loading=True)),
                                            ('moco_free', 'moco', 2, 'none', dict(block_encoder_loading=True)),
                                            ('supmoco_preq', 'supmoco', 3, 'pre_q', dict(pre_trained_encoder_weights=ckpt, data_type='noise',
                                                                                         labelling_strategy='double_precision'))):
        h = define_model('contrastiveblindqrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False,
                         checkpoint_load=False, loss_masking=False, metadata_list=None, lr=1e-3, combined_loss_mode=mode, crop_count=crops,
                         encoder_train_eval='train', encoder_freeze_mode=freeze, **extra, **JOINT_KW)
        og = O.build_oracle('qrcan', num_metadata=256, **JOINT_KW)
        h.net.G.load_state_dict(O.seeded_state_dict(og, 2700))
        if mode == 'moco':
            seed_net(h.net.E, 2710)
        d[tag + '.keys'] = np.array(list(h.net.state_dict().keys()))
        d[tag + '.trainable'] = np.array([k for k, p in h.net.named_parameters() if p.requires_grad])
        for step in range(2):
            x, y = CO.joint_batch(2720 + step, 4, crops)
            kw = dict(metadata=torch.from_numpy(meta), metadata_keys=NOISE_KEYS) if mode == 'supmoco' else {}
            pkg, logits = h.run_train(x=x, y=y, **kw)
            for k, v in pkg.items():
                d['%s.%s%d' % (tag, k, step)] = np.asarray(v)
            d['%s.logits%d' % (tag, step)] = logits.numpy()[:, :48].copy()
            if step == 0:
                for k, p in h.net.named_parameters():
                    if p.requires_grad and p.grad is not None:
                        g = p.grad.detach().numpy().reshape(-1)
                        d['%s.gnorm.%s' % (tag, k)] = np.asarray(np.linalg.norm(g.astype(np.float64)))
                        d['%s.gsample.%s' % (tag, k)] = g[::(29 if k.startswith('G.') else 211)].copy()
        for k, v in h.net.state_dict().items():
            if not k.startswith('E.queue'):
                d['%s.after2.%s' % (tag, k)] = v.detach().numpy().reshape(-1)[::(29 if k.startswith('G.') else 211)].copy()
        d[tag + '.queue_head'] = h.net.E.queue[:, :12].numpy().copy()
        xe, ye = CO.joint_batch(2790, 2, 1)
        ev, evl, _ = h.run_eval(x=xe[:, 0], y=ye[:, 0], request_loss=True)
        d[tag + '.eval_out'] = ev.numpy()[:, :, ::3, ::3].copy()
        d[tag + '.eval_loss'] = np.asarray(evl)
        print(tag, {k: float(v) for k, v in pkg.items()}, 'trainable', len(d[tag + '.trainable']), 'eval loss', float(evl))
    # ---- the 'q' embedding (contrastive_blind_sr.py:137-139: the mlp head's output drives the generator), SR loss only ----
    h = define_model('contrastiveblindqrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False, checkpoint_load=False,
                     loss_masking=False, metadata_list=None, lr=1e-3, block_encoder_loading=True, embedding_type='q', **JOINT_KW)
    opipe = O.build_oracle('contrastiveblindqrcan', **JOINT_KW)
    h.net.load_state_dict(O.seeded_pipeline_state(opipe, 2900))
    for step in range(2):
        x, y = CO.joint_batch(2910 + step, 3, 1)
        loss, out = h.run_train(x=x[:, 0], y=y[:, 0])
        d['qemb.loss%d' % step] = np.asarray(loss)
        if step == 0:
            d['qemb.out0'] = out.numpy()[:, :, ::3, ::3].copy()
    np.savez_compressed(os.path.join(HERE, 'g21_blind_joint_train.npz'), **d)


if __name__ == '__main__' and 'joint' in sys.argv[1:]:
    main_joint()
