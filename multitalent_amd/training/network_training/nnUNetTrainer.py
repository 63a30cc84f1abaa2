"""Trainer base classes — host orchestration around the HIP hot loop (boundary seam 1).

Restates, minimally and in its own words, what the drivers call on a trainer in the reference
(network_trainer.py, nnUNetTrainer.py, nnUNetTrainerV2.py, nnUNetTrainerV2_DDP.py): constructor signature, plans
parsing, network / optimizer construction, poly learning rate, `run_iteration`, checkpoint save/load in the reference's
file format (`.model` = torch.save(dict), `.model.pkl` = {'init','name','class','plans'}), and
`predict_preprocessed_data_return_seg_and_softmax`, `preprocess_patient`, `validate` and the epoch-end bookkeeping that
writes `model_best.model`.  Plotting and the postprocessing search are out of scope.  A trainer consumes any
generator yielding {'data','target','properties'} batches; without one it uses device-resident synthetic batches like the
reference's dummyLoad benchmarking trainer.
"""
import os
import pickle
import sys
import time
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from ... import plans as plans_mod
from ...network_architecture.generic_UNet import Generic_UNet
from ...network_architecture.initialization import InitWeights_He
from ...utilities.nd_softmax import softmax_helper
from ..ds_weights import ds_loss_weights
from ..hot_loop import FusedTrainStep
from ..loss_functions.fused_losses import DC_and_CE_DS_loss


def poly_lr(epoch, max_epochs, initial_lr, exponent=0.9):
    """training/learning_rate/poly_lr.py:16-17."""
    return initial_lr * (1 - epoch / max_epochs) ** exponent


class nnUNetTrainer(object):
    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False):
        self.fp16 = fp16
        self.unpack_data, self.deterministic = unpack_data, deterministic
        self.init_args = (plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data, deterministic, fp16)
        self.stage = stage
        self.plans_file = plans_file
        self.output_folder = output_folder
        self.dataset_directory = dataset_directory
        self.fold = fold
        self.plans = None
        self.batch_dice = batch_dice
        self.network = None
        self.optimizer = None
        self.lr_scheduler = None
        self.amp_grad_scaler = None
        self.tr_gen = self.val_gen = None
        self.was_initialized = False
        self.epoch = 0
        self.max_num_epochs = 1000
        self.num_batches_per_epoch = 250          # network_trainer.py:96-97
        self.num_val_batches_per_epoch = 50
        self.initial_lr = 1e-2                    # nnUNetTrainerV2.py:49-50
        self.weight_decay = 3e-5
        self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics = [], [], [], []
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []      # nnUNetTrainer.py:108-111
        self.best_epoch_based_on_MA_tr_loss = self.best_MA_tr_loss_for_patience = self.best_val_eval_criterion_MA = None
        self.save_every = 50
        # network_trainer.py:73-135 / nnUNetTrainer.py:113-117: moving averages, patience, what gets saved
        self.train_loss_MA_alpha, self.train_loss_MA_eps, self.val_eval_criterion_alpha = 0.93, 5e-4, 0.9
        self.patience, self.lr_threshold = 50, 1e-6
        self.train_loss_MA = self.val_eval_criterion_MA = None
        self.save_latest_only = self.save_intermediate_checkpoints = self.save_best_checkpoint = self.save_final_checkpoint = True
        self.also_val_in_tr_mode = False
        self.dataset = self.dataset_tr = self.dataset_val = None
        self.output_folder_base = output_folder
        self.experiment_name = self.__class__.__name__
        self.gt_niftis_folder = os.path.join(dataset_directory, "gt_segmentations") if dataset_directory is not None else None
        self.log_file = None
        self.regions_class_order = None
        self.classes = self.num_classes = self.patch_size = self.batch_size = None
        self.inference_pad_border_mode = "constant"
        self.inference_pad_kwargs = {'constant_values': 0}
        self.data_aug_params = {'do_mirror': True, 'mirror_axes': (0, 1, 2)}
        self.local_rank = 0
        self.train_step = None
        self.update_fold(fold)

    def update_fold(self, fold):
        """nnUNetTrainer.py:134-152: checkpoints and validation output live in <output_folder>/fold_<k> (or /all)."""
        if fold is None or self.output_folder is None:
            self.fold = fold if fold is not None else self.fold
            return
        name = "%s" % str(fold) if isinstance(fold, str) else "fold_%s" % str(fold)
        if isinstance(fold, str):
            assert fold == "all", "if self.fold is a string then it must be 'all'"
        old = "%s" % str(self.fold) if isinstance(self.fold, str) else "fold_%s" % str(self.fold)
        if self.output_folder.endswith(old):
            self.output_folder = self.output_folder_base
        self.output_folder = os.path.join(self.output_folder, name)
        self.fold = fold

    # ---- plans (nnUNetTrainer.py:319-392) ----------------------------------------------------------------
    def load_plans_file(self):
        self.plans = self.plans_file if isinstance(self.plans_file, dict) else plans_mod.load_plans_file(self.plans_file)

    def process_plans(self, plans):
        if self.stage is None:
            assert len(plans['plans_per_stage']) == 1, "several stages in the plans: pass `stage`"
            self.stage = list(plans['plans_per_stage'].keys())[0]
        self.plans = plans
        sp = plans['plans_per_stage'][self.stage]
        self.batch_size = sp['batch_size']
        self.patch_size = np.array(sp['patch_size']).astype(int)
        self.do_dummy_2D_aug = sp.get('do_dummy_2D_data_aug', False)
        self.net_num_pool_op_kernel_sizes = sp['pool_op_kernel_sizes']
        self.net_conv_kernel_sizes = sp['conv_kernel_sizes']
        self.intensity_properties = plans['dataset_properties']['intensityproperties']
        self.normalization_schemes = plans['normalization_schemes']
        self.base_num_features = plans['base_num_features']
        self.num_input_channels = plans['num_modalities']
        self.num_classes = plans['num_classes'] + 1           # background is added here (nnUNetTrainer.py:366)
        self.classes = plans['all_classes']
        self.use_mask_for_norm = plans['use_mask_for_norm']
        self.transpose_forward = plans.get('transpose_forward', [0, 1, 2])
        self.transpose_backward = plans.get('transpose_backward', [0, 1, 2])
        if len(self.patch_size) != 3:
            raise RuntimeError("only 3D patches are on the hot path (got %s)" % str(self.patch_size))
        self.threeD = True
        self.conv_per_stage = plans.get('conv_per_stage', 2)

    # ---- logging (network_trainer.py:222-254), rank 0 only in DDP -----------------------------------------
    def print_to_log_file(self, *args, also_print_to_console=True, add_timestamp=True):
        if self.local_rank != 0:
            return
        ts = datetime.now()
        if add_timestamp:
            args = ("%s:" % ts, *args)
        if self.output_folder is not None:
            os.makedirs(self.output_folder, exist_ok=True)
            if self.log_file is None:
                self.log_file = os.path.join(self.output_folder, "training_log_%d_%d_%d_%02.0d_%02.0d_%02.0d.txt" %
                                             (ts.year, ts.month, ts.day, ts.hour, ts.minute, ts.second))
            with open(self.log_file, 'a+') as f:
                f.write(" ".join(str(a) for a in args) + "\n")
        if also_print_to_console:
            print(*args)

    # ---- checkpoints (network_trainer.py:256-362, nnUNetTrainer.py:726-734) --------------------------------
    def _optimizer_state_dict(self):
        """torch.optim.SGD-format state built from the flat momentum buffer of the fused step."""
        if self.train_step is not None and getattr(self.train_step, 'head_opt', None) is not None:
            return self.train_step.head_opt.state_dict()          # heads-only phase of the fine-tuning trainers: AdamW state
        params = list(self.network.parameters())
        st = {}
        if self.train_step is not None and self.train_step.buf is not None and not self.train_step.first:
            eng = self.network.engine()
            for i, p in enumerate(params):
                o = eng._views[id(p)][0]
                st[i] = {'momentum_buffer': self.train_step.buf[o:o + p.numel()].view(p.shape).detach().cpu().clone()}
        lr = self.optimizer_lr
        return {'state': st, 'param_groups': [{'lr': lr, 'momentum': 0.99, 'dampening': 0, 'weight_decay': self.weight_decay,
                                               'nesterov': True, 'maximize': False, 'foreach': None, 'differentiable': False,
                                               'fused': None, 'initial_lr': self.initial_lr, 'params': list(range(len(params)))}]}

    def restore_optimizer_state(self, osd):
        """Momentum buffers of a torch.optim.SGD state_dict -> the flat momentum buffer of the fused step."""
        if not torch.cuda.is_available() or not osd.get('state'):
            return
        if getattr(self.train_step, 'head_opt', None) is not None:
            if any('exp_avg' in s for s in osd['state'].values()):
                self.train_step.head_opt.load_state_dict(osd)
            return
        eng = self.network.engine()
        eng.attach(torch.device('cuda', torch.cuda.current_device()))
        self.train_step._state(eng.flat.device)
        params = list(self.network.parameters())
        for i, p in enumerate(params):
            st = osd['state'].get(i)
            if st is not None and st.get('momentum_buffer') is not None:
                o = eng._views[id(p)][0]
                self.train_step.buf[o:o + p.numel()].copy_(st['momentum_buffer'].reshape(-1))
                self.train_step.first = False

    def save_checkpoint(self, fname, save_optimizer=True):
        if self.local_rank != 0:
            return
        sd = {k: v.detach().cpu().clone() for k, v in self.network.state_dict().items()}
        save_this = {'epoch': self.epoch + 1, 'state_dict': sd,
                     'optimizer_state_dict': self._optimizer_state_dict() if save_optimizer else None,
                     'lr_scheduler_state_dict': None,
                     'plot_stuff': (self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics),
                     'best_stuff': (self.best_epoch_based_on_MA_tr_loss, self.best_MA_tr_loss_for_patience, self.best_val_eval_criterion_MA)}
        os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
        torch.save(save_this, fname)
        info = {'init': self.init_args, 'name': self.__class__.__name__, 'class': str(self.__class__), 'plans': self.plans}
        with open(fname + ".pkl", 'wb') as f:
            pickle.dump(info, f)

    def load_checkpoint(self, fname, train=True):
        if not self.was_initialized:
            self.initialize(train)
        self.load_checkpoint_ram(torch.load(fname, map_location='cpu', weights_only=False), train)

    def load_checkpoint_ram(self, checkpoint, train=True):
        """network_trainer.py:331-385 / nnUNetTrainerV2_DDP.py:636-697: strips a leading 'module.' (DDP) from the keys."""
        if not self.was_initialized:
            self.initialize(train)
        cur = self.network.state_dict()
        new = {}
        for k, v in checkpoint['state_dict'].items():
            key = k
            if key not in cur and key.startswith('module.'):
                key = key[7:]
            new[key] = v
        self.network.load_state_dict(new)
        self.network.engine().mark_params_dirty()
        self.epoch = checkpoint['epoch']
        if train and checkpoint.get('optimizer_state_dict') is not None and self.train_step is not None:
            self.restore_optimizer_state(checkpoint['optimizer_state_dict'])
        if 'plot_stuff' in checkpoint:
            self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics = checkpoint['plot_stuff']
        if 'best_stuff' in checkpoint:
            self.best_epoch_based_on_MA_tr_loss, self.best_MA_tr_loss_for_patience, self.best_val_eval_criterion_MA = checkpoint['best_stuff']

    def load_latest_checkpoint(self, train=True):
        for n in ("model_final_checkpoint.model", "model_latest.model", "model_best.model"):
            f = os.path.join(self.output_folder, n)
            if os.path.isfile(f):
                return self.load_checkpoint(f, train=train)
        raise RuntimeError("No checkpoint found")

    def load_best_checkpoint(self, train=True):
        f = os.path.join(self.output_folder, "model_best.model")
        return self.load_checkpoint(f, train) if os.path.isfile(f) else self.load_latest_checkpoint(train)

    def load_final_checkpoint(self, train=False):
        f = os.path.join(self.output_folder, "model_final_checkpoint.model")
        if not os.path.isfile(f):
            raise RuntimeError("Final checkpoint not found. Expected: %s. Please finish the training first." % f)
        return self.load_checkpoint(f, train=train)

    # ---- inference wrapper (nnUNetTrainer.py:483-527, nnUNetTrainerV2_DDP.py:601-634) ------------------------
    def predict_preprocessed_data_return_seg_and_softmax(self, data, do_mirroring=True, mirror_axes=None,
                                                         use_sliding_window=True, step_size=0.5, use_gaussian=True,
                                                         pad_border_mode='constant', pad_kwargs=None, all_in_gpu=False,
                                                         verbose=True, mixed_precision=True, return_device_tensors=False):
        """`return_device_tensors=True` (extension): (seg, probabilities) stay on the device; the probabilities alias the
        network's sliding-window cache until the next prediction."""
        if pad_border_mode == 'constant' and pad_kwargs is None:
            pad_kwargs = {'constant_values': 0}
        if do_mirroring and mirror_axes is None:
            mirror_axes = self.data_aug_params['mirror_axes']
        if do_mirroring:
            assert self.data_aug_params["do_mirror"], "Cannot do mirroring as test time augmentation when training was done without mirroring"
        net = self.network
        ds = self._get_ds(net)
        self._set_ds(net, False)
        was_training = net.training
        net.eval()
        try:
            ret = net.predict_3D(data, do_mirroring=do_mirroring, mirror_axes=mirror_axes, use_sliding_window=use_sliding_window,
                                 step_size=step_size, patch_size=self.patch_size, regions_class_order=self.regions_class_order,
                                 use_gaussian=use_gaussian, pad_border_mode=pad_border_mode, pad_kwargs=pad_kwargs,
                                 all_in_gpu=all_in_gpu, verbose=verbose, mixed_precision=mixed_precision,
                                 return_device_tensors=return_device_tensors)
        finally:
            net.train(was_training)
            self._set_ds(net, ds)
        return ret

    @staticmethod
    def _get_ds(net):
        return net.do_ds

    @staticmethod
    def _set_ds(net, v):
        net.do_ds = v

    # ---- epoch-end bookkeeping (network_trainer.py:509-640) -------------------------------------------------------
    def run_online_evaluation(self, output, target, data=None):
        """nnUNetTrainer.py:683-705 (nnUNetTrainerV2.py:219-223 hands it the full-resolution output and target): hard tp / fp / fn per
        foreground class of argmax(softmax(output)) against the label map, summed over the batch, appended to online_eval_tp / fp / fn
        (+ the per-batch foreground Dice).  Counting is exact integer arithmetic on the device (torch glue: validation only)."""
        with torch.no_grad():
            if output is None:
                x = self.network.engine().forward(data, need_grad=False, all_heads=True)[0]               # NDHWC logits
                seg = x.argmax(-1)                                                                        # softmax is monotone
                nc = x.shape[-1]
            else:
                o = output[0] if isinstance(output, (list, tuple)) else output
                seg = o.argmax(1)
                nc = o.shape[1]
            t = target[0] if isinstance(target, (list, tuple)) else target
            t = t[:, 0].to(seg.device).long()
            tp = np.zeros(nc - 1); fp = np.zeros(nc - 1); fn = np.zeros(nc - 1)
            for c in range(1, nc):
                ps, ts = seg == c, t == c
                tp[c - 1] = float((ps & ts).sum()); fp[c - 1] = float((ps & ~ts).sum()); fn[c - 1] = float((~ps & ts).sum())
            self.online_eval_foreground_dc.append(list((2 * tp) / (2 * tp + fp + fn + 1e-8)))
            self.online_eval_tp.append(list(tp))
            self.online_eval_fp.append(list(fp))
            self.online_eval_fn.append(list(fn))

    def finish_online_evaluation(self):
        """nnUNetTrainer.py:707-728: global foreground Dice per class from the epoch's summed tp / fp / fn (classes never seen dropped),
        its mean appended to all_val_eval_metrics — what update_eval_criterion_MA / model_best selection and the epoch-100 "Dice == 0"
        re-initialisation read.  Without online evaluation this epoch (no validation iterations) the list is left alone and the
        moving average falls back to -validation loss like the reference does for an empty list."""
        if not getattr(self, 'online_eval_tp', None):
            return
        tp, fp, fn = np.sum(self.online_eval_tp, 0), np.sum(self.online_eval_fp, 0), np.sum(self.online_eval_fn, 0)
        with np.errstate(divide='ignore', invalid='ignore'):
            dc = [i for i in [2 * a / (2 * a + b + c) for a, b, c in zip(tp, fp, fn)] if not np.isnan(i)]
        self.all_val_eval_metrics.append(np.mean(dc))
        self.print_to_log_file("Average global foreground Dice:", [np.round(i, 4) for i in dc])
        self.print_to_log_file("(interpret this as an estimate for the Dice of the different classes. This is not exact.)")
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []

    def update_train_loss_MA(self):
        last = self.all_tr_losses[-1]
        self.train_loss_MA = last if self.train_loss_MA is None else \
            self.train_loss_MA_alpha * self.train_loss_MA + (1 - self.train_loss_MA_alpha) * last

    def update_eval_criterion_MA(self):
        """network_trainer.py:527-551: moving average of the validation metric (or of -validation loss)."""
        new = self.all_val_eval_metrics[-1] if len(self.all_val_eval_metrics) > 0 else -self.all_val_losses[-1]
        if self.val_eval_criterion_MA is None:
            self.val_eval_criterion_MA = new
        else:
            a = self.val_eval_criterion_alpha
            self.val_eval_criterion_MA = a * self.val_eval_criterion_MA + (1 - a) * new

    def maybe_save_checkpoint(self):
        """network_trainer.py:509-525."""
        if self.output_folder is None:
            return
        if self.save_intermediate_checkpoints and (self.epoch % self.save_every == (self.save_every - 1)):
            self.print_to_log_file("saving scheduled checkpoint file...")
            if not self.save_latest_only:
                self.save_checkpoint(os.path.join(self.output_folder, "model_ep_%03.0d.model" % (self.epoch + 1)))
            self.save_checkpoint(os.path.join(self.output_folder, "model_latest.model"))
            self.print_to_log_file("done")

    def manage_patience(self):
        """network_trainer.py:553-617: `model_best.model` whenever the validation moving average improves; the patience
        counter on the training-loss moving average (its verdict is ignored by the V2 trainers, which always run all epochs)."""
        if self.patience is None:
            return True
        if self.best_MA_tr_loss_for_patience is None:
            self.best_MA_tr_loss_for_patience = self.train_loss_MA
        if self.best_epoch_based_on_MA_tr_loss is None:
            self.best_epoch_based_on_MA_tr_loss = self.epoch
        if self.best_val_eval_criterion_MA is None:
            self.best_val_eval_criterion_MA = self.val_eval_criterion_MA
        if self.val_eval_criterion_MA > self.best_val_eval_criterion_MA:
            self.best_val_eval_criterion_MA = self.val_eval_criterion_MA
            if self.save_best_checkpoint and self.output_folder is not None:
                self.save_checkpoint(os.path.join(self.output_folder, "model_best.model"))
        if self.train_loss_MA + self.train_loss_MA_eps < self.best_MA_tr_loss_for_patience:
            self.best_MA_tr_loss_for_patience = self.train_loss_MA
            self.best_epoch_based_on_MA_tr_loss = self.epoch
        if self.epoch - self.best_epoch_based_on_MA_tr_loss > self.patience:
            if getattr(self, 'optimizer_lr', self.initial_lr) > self.lr_threshold:
                self.best_epoch_based_on_MA_tr_loss = self.epoch - self.patience // 2
            else:
                return False
        return True

    # ---- unseen data (nnUNetTrainer.py:417-443) -------------------------------------------------------------------
    def preprocess_patient(self, input_files, return_device=False):
        """Read + crop the case on the host, resample to the plan's spacing and normalise on the device.  Returns
        (data [C, X, Y, Z] float32, seg, properties) like the reference (numpy; `return_device=True` keeps the volume in HBM)."""
        name = self.plans.get('preprocessor_name') or "GenericPreprocessor"
        if name != "GenericPreprocessor":
            raise NotImplementedError("preprocessor %s is not on this path (3D GenericPreprocessor only)" % name)
        from ...preprocessing.preprocessing import GenericPreprocessor
        print("using preprocessor", name)
        pre = GenericPreprocessor(self.normalization_schemes, self.use_mask_for_norm, self.transpose_forward, self.intensity_properties)
        return pre.preprocess_test_case(input_files, self.plans['plans_per_stage'][self.stage]['current_spacing'],
                                        return_device=return_device)

    # ---- validation on the held-out cases (nnUNetTrainer.py:526-674) --------------------------------------------------
    def _export_params(self, segmentation_export_kwargs):
        if segmentation_export_kwargs is None:
            p = self.plans.get('segmentation_export_params')
            if p is not None:
                return p['force_separate_z'], p['interpolation_order'], p['interpolation_order_z']
            return None, 1, 0
        k = segmentation_export_kwargs
        return k['force_separate_z'], k['interpolation_order'], k['interpolation_order_z']

    def _validation_world(self):
        """(rank, world): cases are strided over the ranks of a DDP trainer (nnUNetTrainerV2_DDP.py:476)."""
        if getattr(self, 'ddp', False) and dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _validation_barrier(self):
        if getattr(self, 'ddp', False) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()

    def _open_validation(self, do_mirroring, validation_folder_name, args):
        import json
        assert self.was_initialized, "must initialize, ideally with checkpoint (or train first)"
        if self.dataset_val is None:
            self.load_dataset()
            self.do_split()
        out = os.path.join(self.output_folder, validation_folder_name)
        os.makedirs(out, exist_ok=True)
        if self.local_rank == 0:
            with open(os.path.join(out, "validation_args.json"), 'w') as f:
                json.dump(args, f, sort_keys=True, indent=4)
        if do_mirroring:
            if not self.data_aug_params['do_mirror']:
                raise RuntimeError("We did not train with mirroring so you cannot do inference with mirroring enabled")
            return out, self.data_aug_params['mirror_axes']
        return out, ()

    def _predict_validation_case(self, k, do_mirroring, mirror_axes, use_sliding_window, step_size, use_gaussian, all_in_gpu):
        """probabilities of one preprocessed case on the device, transposed back ([C, *original axis order])."""
        data = np.load(self.dataset[k]['data_file'])['data']
        print(k, data.shape)
        net = self.network
        ds = self._get_ds(net)
        self._set_ds(net, False)
        try:
            probs = net.predict_3D(np.ascontiguousarray(data[:-1]), do_mirroring=do_mirroring, mirror_axes=mirror_axes,
                                   use_sliding_window=use_sliding_window, step_size=step_size, patch_size=self.patch_size,
                                   regions_class_order=self.regions_class_order, use_gaussian=use_gaussian,
                                   pad_border_mode='constant', pad_kwargs={'constant_values': 0}, all_in_gpu=all_in_gpu,
                                   verbose=False, mixed_precision=self.fp16, return_device_tensors=True)[1]
        finally:
            self._set_ds(net, ds)
        if list(self.transpose_backward) != [0, 1, 2]:
            probs = probs.permute(0, *[int(i) + 1 for i in self.transpose_backward]).contiguous()
        return probs

    def validate(self, do_mirroring=True, use_sliding_window=True, step_size=0.5, save_softmax=True, use_gaussian=True,
                 overwrite=True, validation_folder_name='validation_raw', debug=False, all_in_gpu=False,
                 segmentation_export_kwargs=None, run_postprocessing_on_folds=True):
        """nnUNetTrainer.py:526-674 (DDP: nnUNetTrainerV2_DDP.py:431-599): every validation case through the sliding window
        and the device export (probabilities never leave HBM), `summary.json` from `aggregate_scores` when the ground-truth
        folder exists.  The connected-component postprocessing search (`run_postprocessing_on_folds`) is not on this path."""
        import pickle
        from ...evaluation.evaluator import aggregate_scores
        from ...inference.segmentation_export import save_segmentation_nifti_from_softmax
        current_mode = self.network.training
        self.network.eval()
        args = {'do_mirroring': do_mirroring, 'use_sliding_window': use_sliding_window, 'step_size': step_size,
                'save_softmax': save_softmax, 'use_gaussian': use_gaussian, 'overwrite': overwrite,
                'validation_folder_name': validation_folder_name, 'debug': debug, 'all_in_gpu': all_in_gpu,
                'segmentation_export_kwargs': segmentation_export_kwargs}
        out, mirror_axes = self._open_validation(do_mirroring, validation_folder_name, args)
        force_separate_z, order, order_z = self._export_params(segmentation_export_kwargs)
        rank, world = self._validation_world()
        all_keys = list(self.dataset_val.keys())
        my_keys = all_keys[rank::world]
        pred_gt_tuples = []
        for k in all_keys:
            with open(self.dataset[k]['properties_file'], 'rb') as f:
                properties = pickle.load(f)
            fname = properties['list_of_data_files'][0].split("/")[-1][:-12]
            pred_gt_tuples.append([os.path.join(out, fname + ".nii.gz"), os.path.join(self.gt_niftis_folder or '', fname + ".nii.gz")])
            if k not in my_keys:
                continue
            if overwrite or not os.path.isfile(os.path.join(out, fname + ".nii.gz")) or \
                    (save_softmax and not os.path.isfile(os.path.join(out, fname + ".npz"))):
                probs = self._predict_validation_case(k, do_mirroring, mirror_axes, use_sliding_window, step_size, use_gaussian,
                                                      all_in_gpu)
                save_segmentation_nifti_from_softmax(probs, os.path.join(out, fname + ".nii.gz"), properties, order,
                                                     self.regions_class_order, None, None,
                                                     os.path.join(out, fname + ".npz") if save_softmax else None, None,
                                                     force_separate_z, order_z, verbose=False)
        self._validation_barrier()
        self.print_to_log_file("finished prediction")
        if rank == 0 and self.gt_niftis_folder is not None and os.path.isdir(self.gt_niftis_folder):
            self.print_to_log_file("evaluation of raw predictions")
            aggregate_scores(pred_gt_tuples, labels=list(range(self.num_classes)), json_output_file=os.path.join(out, "summary.json"),
                             json_name=self.experiment_name + " val tiled %s" % str(use_sliding_window), json_author="Fabian",
                             json_task=(self.dataset_directory or "").split("/")[-1])
        self.network.train(current_mode)
        self._validation_barrier()


class nnUNetTrainerV2(nnUNetTrainer):
    """nnUNetTrainerV2.py:39-444 (single GPU): Generic_UNet + deep supervision + SGD-Nesterov + poly lr."""

    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False):
        super().__init__(plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data, deterministic, fp16)
        self.max_num_epochs = 1000
        self.initial_lr = 1e-2
        self.deep_supervision_scales = None
        self.ds_loss_weights = None
        self.pin_memory = True
        self.optimizer_lr = self.initial_lr
        self.ddp = False

    def setup_DA_params(self):
        """only what the hot path needs: the deep-supervision target scales (nnUNetTrainerV2.py:107-108)."""
        self.deep_supervision_scales = [[1, 1, 1]] + list(list(i) for i in 1 / np.cumprod(
            np.vstack(self.net_num_pool_op_kernel_sizes), axis=0))[:-1]

    def make_loss(self):
        return DC_and_CE_DS_loss(self.ds_loss_weights, batch_dice=self.batch_dice, smooth=1e-5, do_bg=False, ddp=self.ddp)

    def initialize(self, training=True, force_load_plans=False):
        if self.was_initialized:
            return
        if force_load_plans or self.plans is None:
            self.load_plans_file()
        self.process_plans(self.plans)
        self.setup_DA_params()
        self.ds_loss_weights = ds_loss_weights(len(self.net_num_pool_op_kernel_sizes))    # nnUNetTrainerV2.py:78-90
        self.initialize_network()
        # fp16=True is the reference's mixed-precision switch (autocast + GradScaler, nnUNetTrainerV2.py:236-249); here it
        # selects bf16 matrix inputs with fp32 accumulation (Engine.set_precision) — no loss scaling needed
        self.network.engine().set_precision('bf16' if self.fp16 else 'fp32')
        self.initialize_optimizer_and_scheduler()
        self.was_initialized = True

    def initialize_network(self):
        """nnUNetTrainerV2.py:131-164 — same positional call."""
        self.network = Generic_UNet(self.num_input_channels, self.base_num_features, self.num_classes,
                                    len(self.net_num_pool_op_kernel_sizes), self.conv_per_stage, 2, nn.Conv3d, nn.InstanceNorm3d,
                                    {'eps': 1e-5, 'affine': True}, nn.Dropout3d, {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                    {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x, InitWeights_He(1e-2),
                                    self.net_num_pool_op_kernel_sizes, self.net_conv_kernel_sizes, False, True, True)
        if torch.cuda.is_available():
            self.network.cuda()
        self.network.inference_apply_nonlin = softmax_helper

    def initialize_optimizer_and_scheduler(self):
        """SGD(lr, weight_decay 3e-5, momentum 0.99, nesterov) (nnUNetTrainerV2.py:166-170) as ONE fused kernel."""
        assert self.network is not None
        self.train_step = FusedTrainStep(self.network, self.make_loss(), lr=self.initial_lr, weight_decay=self.weight_decay,
                                         momentum=0.99, max_norm=12.0, ddp=self.ddp)
        self.optimizer = self.train_step          # drivers only touch .param_groups-like lr through maybe_update_lr
        self.lr_scheduler = None

    def maybe_update_lr(self, epoch=None):
        ep = self.epoch + 1 if epoch is None else epoch                                  # nnUNetTrainerV2.py:393-408
        self.optimizer_lr = poly_lr(ep, self.max_num_epochs, self.initial_lr, 0.9)
        self.train_step.lr = self.optimizer_lr
        self.print_to_log_file("lr:", np.round(self.optimizer_lr, decimals=6))

    def _to_device(self, a):
        if isinstance(a, (list, tuple)):
            return [self._to_device(i) for i in a]
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(a).float()
        return a.cuda(non_blocking=True) if not a.is_cuda else a

    def prepare_target(self, target):
        """The reference's augmenter delivers the deep-supervision pyramid as a list (DownsampleSegForDSTransform2 in CPU
        workers).  A single full-resolution label map [B,1,D,H,W] is also accepted: the pyramid is then built on the device
        (RemoveLabelTransform(-1, 0) included)."""
        target = self._to_device(target)
        if torch.is_tensor(target):
            from ..data_augmentation.downsampling import downsample_seg_for_ds_transform2
            target = downsample_seg_for_ds_transform2(target, self.deep_supervision_scales, 0, None, remove_minus_one=True)
        return target

    def loss_args(self, data_dict):
        return (self.prepare_target(data_dict['target']),)

    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False):
        """nnUNetTrainerV2.py:225-274: fwd, loss, bwd, clip 12, step — one call into the fused hot loop."""
        data_dict = next(data_generator)
        data = self._to_device(data_dict['data'])
        largs = self.loss_args(data_dict)
        res = self.train_step(data, *largs, do_backprop=do_backprop)
        l = res[0] if isinstance(res, tuple) else res
        if run_online_evaluation:
            # the output of the SAME forward pass (nnUNetTrainerV2.py:256-258), not a second one
            self.run_online_evaluation([self.train_step.last_logits], largs[0])
        return l.detach().cpu().numpy()

    def on_epoch_end(self):
        """network_trainer.py:619-633 followed by nnUNetTrainerV2.py:410-430: the V2 trainers ignore the patience verdict and
        run all epochs; at epoch 100 a validation Dice of exactly 0 lowers the momentum to 0.95 and re-initialises the weights."""
        self.finish_online_evaluation()
        self.maybe_update_lr()
        self.maybe_save_checkpoint()
        if len(self.all_val_eval_metrics) > 0 or len(self.all_val_losses) > 0:       # nothing to average before the first validation
            self.update_eval_criterion_MA()
            if self.train_loss_MA is not None:
                self.manage_patience()
        if self.epoch == 100 and len(self.all_val_eval_metrics) > 0 and self.all_val_eval_metrics[-1] == 0:
            self.train_step.mom = 0.95
            self.network.apply(InitWeights_He(1e-2))
            self.network.engine().mark_params_dirty()
            self.print_to_log_file("At epoch 100, the mean foreground Dice was 0: momentum reduced to 0.95, weights reinitialized")
        return self.epoch < self.max_num_epochs

    def _default_generator(self):
        from ...synthetic import SyntheticBatchGenerator
        return SyntheticBatchGenerator(self)

    # ---- real data: preprocessed cases on disk ---------------------------------------------
    @property
    def folder_with_preprocessed_data(self):
        """<dataset_directory>/<data_identifier>_stage<stage> (nnUNetTrainer.py:389-391 / :181-182)."""
        if self.dataset_directory is None or self.plans is None:
            return None
        return os.path.join(self.dataset_directory, self.plans['data_identifier'] + "_stage%d" % self.stage)

    def load_dataset(self):
        from ..dataloading.dataset_loading import load_dataset
        self.dataset = load_dataset(self.folder_with_preprocessed_data)              # nnUNetTrainer.py:411-412

    def do_split(self):
        """nnUNetTrainerV2.py:276-340: splits_final.pkl (created as a seeded 5-fold split when missing), fold 'all', or a seeded
        80:20 split for folds beyond the file."""
        import pickle
        from collections import OrderedDict
        if self.fold == "all":
            tr_keys = val_keys = list(self.dataset.keys())
        else:
            splits_file = os.path.join(self.dataset_directory, "splits_final.pkl")
            if not os.path.isfile(splits_file):
                from sklearn.model_selection import KFold
                self.print_to_log_file("Creating new 5-fold cross-validation split...")
                splits = []
                all_keys_sorted = np.sort(list(self.dataset.keys()))
                for train_idx, test_idx in KFold(n_splits=5, shuffle=True, random_state=12345).split(all_keys_sorted):
                    splits.append(OrderedDict())
                    splits[-1]['train'] = np.array(all_keys_sorted)[train_idx]
                    splits[-1]['val'] = np.array(all_keys_sorted)[test_idx]
                if self.local_rank == 0:             # one writer, atomic rename: other ranks never read a partial file
                    with open(splits_file + ".tmp", 'wb') as f:
                        pickle.dump(splits, f)
                    os.replace(splits_file + ".tmp", splits_file)
            else:
                with open(splits_file, 'rb') as f:
                    splits = pickle.load(f)
            if self.fold < len(splits):
                tr_keys, val_keys = splits[self.fold]['train'], splits[self.fold]['val']
            else:
                rnd = np.random.RandomState(seed=12345 + self.fold)
                keys = np.sort(list(self.dataset.keys()))
                idx_tr = rnd.choice(len(keys), int(len(keys) * 0.8), replace=False)
                tr_keys = [keys[i] for i in idx_tr]
                val_keys = [keys[i] for i in range(len(keys)) if i not in idx_tr]
        tr_keys, val_keys = sorted(tr_keys), sorted(val_keys)
        self.dataset_tr = OrderedDict((i, self.dataset[i]) for i in tr_keys)
        self.dataset_val = OrderedDict((i, self.dataset[i]) for i in val_keys)

    oversample_foreground_percent = 0.33                                            # nnUNetTrainer.py:131
    pad_all_sides = None

    def _sampling_probabilities(self, keys):
        return None

    def get_basic_generators(self):
        """nnUNetTrainer.py:394-409 (MultiTalent: …_Trainer_DDP.py:625-661).  The loader draws network-sized patches: the
        oversized `basic_generator_patch_size` only exists for the CPU SpatialTransform, which is not part of this path."""
        from ..dataloading.dataset_loading import DataLoader3D
        self.load_dataset()
        self.do_split()
        ps = tuple(int(i) for i in self.patch_size)
        bps = tuple(int(i) for i in self.basic_generator_patch_size) if self.device_augmentation else ps
        mk = lambda ds, p: DataLoader3D(ds, p, ps, self.batch_size, False, oversample_foreground_percent=self.oversample_foreground_percent,
                                        pad_mode="constant", pad_sides=self.pad_all_sides, memmap_mode='r',
                                        sampling_probabilities=self._sampling_probabilities(list(ds.keys())))
        return mk(self.dataset_tr, bps), mk(self.dataset_val, ps)

    device_augmentation = True      # rotation/scaling/intensity/mirror augmentation of the training batches on the device

    def setup_augmentation_params(self):
        """nnUNetTrainerV2.setup_DA_params (:352-389): +-30 degree rotations (+-180 in-plane for the dummy-2D mode of anisotropic
        patches), scale (0.7, 1.4), no elastic deformation; the loader patch is computed BEFORE the scale override, i.e. with
        the default (0.85, 1.25) — a quirk of the reference that is kept."""
        from ..data_augmentation.color import default_3d_augmentation_params
        from ..data_augmentation.spatial import get_patch_size
        p = default_3d_augmentation_params()
        ps = [int(i) for i in self.patch_size]
        if getattr(self, 'do_dummy_2D_aug', False):
            p['dummy_2D'] = True
            p['rotation_x'] = (-np.pi, np.pi)                      # default_2D_augmentation_params['rotation_x']
            self.basic_generator_patch_size = np.array([ps[0]] + list(get_patch_size(ps[1:], p['rotation_x'], p['rotation_y'],
                                                                                     p['rotation_z'], (0.85, 1.25))))
        else:
            self.basic_generator_patch_size = get_patch_size(ps, p['rotation_x'], p['rotation_y'], p['rotation_z'], (0.85, 1.25))
        self.data_aug_params.update(p)
        return p

    def maybe_setup_data_generators(self):
        """real cases when <dataset_directory>/<data_identifier>_stage<k> exists and no generator was attached."""
        f = self.folder_with_preprocessed_data
        if self.tr_gen is None and f is not None and os.path.isdir(f):
            self.setup_data_generators()

    def setup_data_generators(self):
        """tr_gen / val_gen from the preprocessed cases under dataset_directory (unpacked to .npy first, like
        nnUNetTrainerV2.initialize :113-121); the batches carry ONE label map, the device builds the pyramid."""
        from ..dataloading.dataset_loading import SegToTargetGenerator, unpack_dataset
        if self.unpack_data and self.local_rank == 0:
            unpack_dataset(self.folder_with_preprocessed_data)
        if self.ddp and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        params = self.setup_augmentation_params() if self.device_augmentation else None
        dl_tr, dl_val = self.get_basic_generators()
        self.val_gen = SegToTargetGenerator(dl_val)
        if self.device_augmentation and torch.cuda.is_available():
            from ..data_augmentation.color import MoreDADeviceAugmenter
            self.tr_gen = MoreDADeviceAugmenter(dl_tr, tuple(int(i) for i in self.patch_size), params,
                                                torch.device('cuda', torch.cuda.current_device()))
        else:
            self.tr_gen = SegToTargetGenerator(dl_tr, tuple(int(i) for i in self.patch_size))

    def _log_epoch_losses(self, tr, va):
        """tr / va: per-iteration return values of run_iteration (scalars here, (loss, ce, dc) for the MultiTalent trainers)."""
        tr, va = np.asarray(tr, dtype=np.float64), np.asarray(va, dtype=np.float64)
        self.all_tr_losses.append(float(tr.mean()))
        self.print_to_log_file("train loss : %.4f" % self.all_tr_losses[-1])
        self.all_val_losses.append(float(va.mean()))
        self.print_to_log_file("validation loss: %.4f" % self.all_val_losses[-1])

    def run_training(self):
        """epoch loop of network_trainer.py:411-505 (with nnUNetTrainerV2.run_training's lr reset) without plotting: train
        iterations, validation iterations with online evaluation, moving averages, checkpoints (latest / best / final)."""
        if not self.was_initialized:
            self.initialize(True)
        self.maybe_update_lr(self.epoch)
        self.maybe_setup_data_generators()
        if self.tr_gen is None:
            self.tr_gen = self._default_generator()
        if self.val_gen is None:
            self.val_gen = self.tr_gen
        if self.output_folder is not None and self.local_rank == 0:
            os.makedirs(self.output_folder, exist_ok=True)
        while self.epoch < self.max_num_epochs:
            self.print_to_log_file("\nepoch: ", self.epoch)
            t0 = time.time()
            self.network.train()
            tr = [self.run_iteration(self.tr_gen, True) for _ in range(self.num_batches_per_epoch)]
            with torch.no_grad():
                self.network.eval()
                va = [self.run_iteration(self.val_gen, False, True) for _ in range(self.num_val_batches_per_epoch)]
            self._log_epoch_losses(tr, va)
            self.update_train_loss_MA()
            cont = self.on_epoch_end()
            if not cont:
                break
            self.epoch += 1
            self.print_to_log_file("This epoch took %f s\n" % (time.time() - t0))
        self.epoch -= 1             # network_trainer.py:493: the final checkpoint stores epoch + 1 == max_num_epochs
        if self.output_folder is not None:
            if self.save_final_checkpoint:
                self.save_checkpoint(os.path.join(self.output_folder, "model_final_checkpoint.model"))
            if self.local_rank == 0:
                for n in ("model_latest.model", "model_latest.model.pkl"):      # identical to the final one
                    f = os.path.join(self.output_folder, n)
                    if os.path.isfile(f):
                        os.remove(f)


class nnUNetTrainerV2_DDP(nnUNetTrainerV2):
    """nnUNetTrainerV2_DDP.py:46-697: one process per GPU, RCCL ('nccl' backend) gradient all-reduce overlapped with
    backward (hot_loop.GradAllReducer) instead of torch DDP buckets."""

    def __init__(self, plans_file, fold, local_rank, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, distribute_batch_size=False, fp16=False):
        super().__init__(plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data, deterministic, fp16)
        self.init_args = (plans_file, fold, local_rank, output_folder, dataset_directory, batch_dice, stage, unpack_data,
                          deterministic, distribute_batch_size, fp16)
        self.distribute_batch_size = distribute_batch_size
        np.random.seed(local_rank)
        torch.manual_seed(local_rank)
        self.local_rank = local_rank
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            torch.cuda.manual_seed_all(local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend='nccl' if torch.cuda.is_available() else 'gloo', init_method='env://')
        self.ddp = True
        self.oversample_foreground_percent = 0.33

    def set_batch_size_and_oversample(self):
        """Per-rank batch size and foreground-oversampling share (nnUNetTrainerV2_DDP.py:75-117): ranks are laid side by
        side on the [0, global batch) sample axis and the last `oversample_foreground_percent` of that axis is the
        foreground-forced part — in --dbs mode the plan batch is split, otherwise every rank gets the plan batch."""
        world, me = dist.get_world_size(), dist.get_rank()
        plan_bs, fg = self.batch_size, self.oversample_foreground_percent
        total = plan_bs if self.distribute_batch_size else plan_bs * world
        share = int(np.ceil(plan_bs / world))
        sizes = []
        for r in range(world):
            sizes.append(min(share, plan_bs - r * share) if self.distribute_batch_size else plan_bs)
        if sizes[me] <= 0:
            raise RuntimeError("--dbs with %d ranks and plan batch size %d leaves rank %d without samples "
                               "(run without distribute_batch_size)" % (world, plan_bs, me))
        lo, hi = float(np.sum(sizes[:me])) / total, float(np.sum(sizes[:me + 1])) / total
        if hi < 1 - fg:
            pct = 0.0
        elif lo > 1 - fg:
            pct = 1.0
        else:
            pct = 1 - ((1 - fg) - lo) / (hi - lo)
        self.global_batch_size = total
        self.batch_size, self.oversample_foreground_percent = int(sizes[me]), float(pct)
        return total

    def process_plans(self, plans):
        super().process_plans(plans)
        self.set_batch_size_and_oversample()

    def initialize_network(self):
        torch.manual_seed(1234)      # identical initial weights on every rank (what DDP's initial broadcast guarantees)
        super().initialize_network()
        if dist.is_initialized() and dist.get_world_size() > 1 and torch.cuda.is_available():
            eng = self.network.engine()
            eng.attach(torch.device('cuda', self.local_rank))
            dist.broadcast(eng.flat, 0)
            eng.mark_params_dirty()
