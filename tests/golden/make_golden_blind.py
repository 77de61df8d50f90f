"""Golden fixture G13 (blind-SR pipeline: frozen contrastive encoder + QRCAN) from the REAL reference handler.

Runs ONLY in the build container (needs /root/reference):   python tests/golden/make_golden_blind.py
define_model('contrastiveblindqrcan', ..., block_encoder_loading=True) - the configuration of the reference's own CPU test
(automated_testing/sisr_tests/test_model_cpu_execute.py:71-88), reduced to 16 features, 2 groups x 2 blocks, x2 - is built on the
CPU, its state set from oracle.sr_oracle.seeded_pipeline_state, and driven through the reference's run_train / run_eval:
three training steps (loss, output and every generator gradient of step 0, generator weights and encoder BatchNorm running statistics
after step 3, learning rates), one evaluation, and the encoder's eval-mode embedding of a seeded batch.
"""
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
from rumpy.shared_framework.models import define_model  # noqa: E402

KW = dict(scale=2, n_feats=16, n_resgroups=2, n_resblocks=2, reduction=16, style='standard', include_q_layer=True,
          selective_meta_blocks=[True, False], num_q_layers_inner_residual=1)


def main():
    torch.manual_seed(0)
    h = define_model('contrastiveblindqrcan', model_save_dir=tempfile.mkdtemp(), device=torch.device('cpu'), eval_mode=False,
                     checkpoint_load=False, loss_masking=False, metadata_list=None, block_encoder_loading=True, lr=1e-3,
                     scheduler='cosine_annealing_warm_restarts', scheduler_params={'t_mult': 1, 'restart_period': 5, 'lr_min': 1e-7}, **KW)
    opipe = O.build_oracle('contrastiveblindqrcan', **KW)
    h.net.load_state_dict(O.seeded_pipeline_state(opipe, 900))
    d = {'keys': np.array(list(h.net.state_dict().keys())),
         'trainable': np.array([k for k, p in h.net.named_parameters() if p.requires_grad]),
         'optimizer_params': np.asarray(len(h.optimizer.param_groups[0]['params']))}
    for step in range(3):
        xb, yb = O.synthetic_batch(910 + step, 3, lr_hw=12, scale=2)
        loss, o = h.run_train(x=xb, y=yb)
        d['loss%d' % step] = np.asarray(loss)
        d['lr_after%d' % step] = np.asarray(h.get_learning_rate())
        d['encoder_training_flag%d' % step] = np.asarray(h.net.E.training)     # True: see OracleBlindPipeline's note
        if step == 0:
            d['out0'] = o.detach().numpy()
            for k, p in h.net.named_parameters():
                if p.requires_grad:
                    d['grad0.' + k] = p.grad.detach().numpy().copy()
    for k, v in h.net.state_dict().items():
        if k.startswith('G.') or 'running' in k or 'num_batches' in k:
            d['w3.' + k] = v.detach().numpy().copy()
    xe, ye = O.synthetic_batch(990, 2, lr_hw=10, scale=2)
    ev, evl, _ = h.run_eval(x=xe, y=ye, request_loss=True)
    d['eval_out'] = ev.detach().numpy()
    d['eval_loss'] = np.asarray(evl)
    with torch.no_grad():
        d['eval_embedding'] = h.net.E(xe)[0].numpy()          # net is in eval mode after run_eval: running statistics
    np.savez_compressed(os.path.join(HERE, 'g13_blind_qrcan_small_train.npz'), **d)
    print('wrote g13; keys', len(d['keys']), 'trainable', len(d['trainable']), 'optimizer params', int(d['optimizer_params']),
          'E.training during run_train', bool(d['encoder_training_flag0']))


if __name__ == '__main__':
    main()
