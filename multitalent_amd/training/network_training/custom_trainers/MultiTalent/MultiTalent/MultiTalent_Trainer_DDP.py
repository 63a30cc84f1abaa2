"""MultiTalent trainers on the HIP engine (plain Generic_UNet variant).

Mirrors MultiTalent_trainer_ddp / MultiTalent_trainer_ddp_2000ep of the reference
(custom_trainers/MultiTalent/MultiTalent/MultiTalent_Trainer_DDP.py:30-127,324-370,544-623,796-808): same class names
(the drivers find trainers by NAME), same constructor signature, num_classes = 47 regions, sigmoid inference
non-linearity, regions_class_order = range(47), per-dataset BCE + Dice loss with cross-rank batch Dice, and
`run_iteration` returning (loss, ce, dc).  `nnUNetTrainerV2_MultiTalent` is the name BASELINE.json's north_star uses;
it is an alias of MultiTalent_trainer_ddp (the reference has no class of that name)."""
import numpy as np
import torch
from torch import nn

from ....nnUNetTrainer import nnUNetTrainerV2_DDP
from .....ds_weights import ds_loss_weights
from .....loss_functions.fused_losses import MultiTalentLoss
from ......dataset_conversion.Task100_MultiTalent import (MultiTalent_region_output_idx_mapping, MultiTalent_regions,
                                                          MultiTalent_valid_regions)


class MultiTalent_trainer_ddp(nnUNetTrainerV2_DDP):
    def __init__(self, plans_file, fold, local_rank, output_folder=None, dataset_directory=None, batch_dice=True,
                 stage=None, unpack_data=True, deterministic=True, distribute_batch_size=False, fp16=False):
        batch_dice = True                                                     # reference :33
        super().__init__(plans_file=plans_file, fold=fold, local_rank=local_rank, output_folder=output_folder,
                         dataset_directory=dataset_directory, batch_dice=batch_dice, stage=stage, unpack_data=unpack_data,
                         deterministic=deterministic, fp16=fp16, distribute_batch_size=distribute_batch_size)
        self.regions = MultiTalent_regions
        self.loss = None
        self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], []

    def process_plans(self, plans):
        super().process_plans(plans)
        self.num_classes = len(self.regions)                                  # reference :48-51

    def make_loss(self):
        return MultiTalentLoss(self.ds_loss_weights, batch_dice=self.batch_dice, regions=self.regions,
                               region_idx=MultiTalent_region_output_idx_mapping)

    def initialize_network(self):
        super().initialize_network()
        self.network.inference_apply_nonlin = nn.Sigmoid()                    # reference :43-46

    def initialize(self, training=True, force_load_plans=False):
        super().initialize(training, force_load_plans)
        self.regions_class_order = list(range(self.num_classes))              # reference :127

    def _sampling_probabilities(self, keys):
        """p(case) ~ 1/sqrt(cases of its dataset) (reference :629-647); the per-dataset totals are logged like the reference."""
        from .....dataloading.dataset_loading import sqrt_sampling_probabilities
        p, per_dataset = sqrt_sampling_probabilities(keys)
        self.dataset_prob = per_dataset
        self.print_to_log_file('probabilities per dataset:', per_dataset)
        return p

    def compute_loss(self, output, target, valid_regions):
        """Signature of the reference's compute_loss (:544-623); fused statistics kernels + [B,47] glue."""
        return self.train_step.loss_fn(output, target, valid_regions)

    def loss_args(self, data_dict):
        valid_regions = [p['valid_regions'] for p in data_dict['properties']]     # reference :329
        return (self.prepare_target(data_dict['target']), valid_regions)

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False):
        data_dict = next(data_generator)
        data = self._to_device(data_dict['data'])
        largs = self.loss_args(data_dict)
        l, ce, dc = self.train_step(data, *largs, do_backprop=do_backprop)
        if run_online_evaluation:
            self.run_online_evaluation(None, largs[0], largs[1], data=data)
        return l.detach().cpu().numpy(), ce.detach().cpu().numpy(), dc.detach().cpu().numpy()

    def run_online_evaluation(self, output, target, valid_regions, data=None):
        """Hard (sigmoid > 0.5) tp/fp/fn per valid region on the full-resolution output (reference :372-410): the fused
        statistics kernel applied to saturated logits gives exactly these counts."""
        from .....distributed_utils import sum_over_ranks
        from .....loss_functions.fused_losses import _MultiTalentStats, _target_flat
        with torch.no_grad():
            if output is None:
                output = [o.permute(0, 4, 1, 2, 3) for o in self.network.engine().forward(data, need_grad=False, all_heads=True)]
            hard = torch.where(output[0] > 0, torch.full_like(output[0], 80.0), torch.full_like(output[0], -80.0))
            valid, lut = self.train_step.loss_fn._masks(valid_regions, hard.device)
            st = _MultiTalentStats.apply(hard, _target_flat(target[0]), valid, lut)[..., 1:]       # [B, C, 3]
            st = st.detach().cpu().numpy()
        self.online_eval_tp.append(list(st[..., 0].sum(0)))
        self.online_eval_fp.append(list(st[..., 1].sum(0)))
        self.online_eval_fn.append(list(st[..., 2].sum(0)))

    def run_training(self):
        """Same epoch structure as the reference (:663-792): 250 train + 50 validation iterations, logging loss/CE/Dice."""
        import os
        import time
        if not self.was_initialized:
            self.initialize(True)
        self.maybe_update_lr(self.epoch)
        self.maybe_setup_data_generators()
        if self.tr_gen is None:
            self.tr_gen = self._default_generator()
        if self.val_gen is None:
            self.val_gen = self.tr_gen
        while self.epoch < self.max_num_epochs:
            t0 = time.time()
            self.network.train()
            tr = np.array([self.run_iteration(self.tr_gen, True) for _ in range(self.num_batches_per_epoch)])
            self.all_tr_losses.append(float(tr[:, 0].mean()))
            self.print_to_log_file("\nepoch: ", self.epoch)
            self.print_to_log_file("train loss : %.4f  ce: %.4f  dice: %.4f" % tuple(tr.mean(0)))
            with torch.no_grad():
                self.network.eval()
                va = np.array([self.run_iteration(self.val_gen, False, True) for _ in range(self.num_val_batches_per_epoch)])
                self.all_val_losses.append(float(va[:, 0].mean()))
            self.print_to_log_file("validation loss: %.4f" % self.all_val_losses[-1], "This epoch took %f s\n" % (time.time() - t0))
            cont = self.on_epoch_end()
            self.epoch += 1
            if not cont:
                break
        if self.output_folder is not None:
            self.save_checkpoint(os.path.join(self.output_folder, "model_final_checkpoint.model"))


class MultiTalent_trainer_ddp_2000ep(MultiTalent_trainer_ddp):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_num_epochs = 2000                                            # reference :796-808


class nnUNetTrainerV2_MultiTalent(MultiTalent_trainer_ddp):
    """Name used by BASELINE.json's north_star; behaviour of MultiTalent_trainer_ddp."""
