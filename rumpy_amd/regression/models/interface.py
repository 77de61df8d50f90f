"""Regression model interface of the MI355X path (image in, vector out) - rumpy/regression/models/interface.py:6-21 over the part of
rumpy/shared_framework/models/base_interface.py:23-315 that builds the handler: the object the reference's own contrastive tests drive
(automated_testing/contrastive_tests/test_contrastive_cpu_execute.py:21-60).  Directory / config bookkeeping is out of scope, as for
rumpy_amd/SISR/models/interface.py."""
import os

from rumpy_amd.shared_framework.configuration.gpu_check import device_selector
from rumpy_amd.shared_framework.models import define_model


class RegressionInterface:
    def __init__(self, model_loc, experiment, gpu='off', sp_gpu=0, mode='eval', new_params=None, load_epoch=None, checkpoint_load=False,
                 no_directories=False, **kwargs):
        """new_params: {'name': ..., 'internal_params': {...}} as in the reference TOML [model] table."""
        self.device = device_selector(gpu, sp_gpu)
        self.mode = mode
        self.metadata = new_params
        self.name = new_params['name'].lower()
        self.base_folder = os.path.join(str(model_loc), experiment)
        self.saved_models = os.path.join(self.base_folder, 'saved_models')
        if not no_directories:
            os.makedirs(self.saved_models, exist_ok=True)
        self.model_epoch = 0
        self.model = define_model(self.name, model_save_dir=self.saved_models, device=self.device, eval_mode=(mode == 'eval'),
                                  checkpoint_load=checkpoint_load, **new_params['internal_params'])
        self.configuration = {'colorspace': self.model.colorspace, 'input': self.model.im_input}
        if load_epoch is not None:
            state = self.model.load_model('train_model', load_epoch, legacy=self.model.legacy_load)
            self.model_epoch = state['model_epoch']
        if gpu == 'multi':
            self.model.set_multi_gpu()

    def train_batch(self, lr, target_metadata, **kwargs):
        """LR image crops in, the batch's degradation metadata (or another regression target) as ground truth (:13-17)"""
        return self.model.run_train(x=lr, y=target_metadata, **kwargs)

    def net_run_and_process(self, lr=None, target_metadata=None, *args, **kwargs):
        out_vector, loss, timing = self.model.run_eval(x=lr, y=target_metadata, **kwargs)
        return out_vector, loss, timing
