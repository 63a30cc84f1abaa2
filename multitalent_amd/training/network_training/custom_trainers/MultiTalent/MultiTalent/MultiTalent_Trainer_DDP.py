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
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []

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

    # ---- folds (reference :433-543) ----------------------------------------------------------------------------------------
    def _preprocessed_root(self):
        """nnunet.paths.preprocessing_output_dir: $nnUNet_preprocessed, else the folder holding this task's dataset_directory."""
        import os
        return os.environ.get('nnUNet_preprocessed') or os.path.dirname(os.path.abspath(self.dataset_directory))

    def _task_folder(self, task_id):
        """convert_id_to_task_name restricted to the preprocessed root (the only place the split files live)."""
        import os
        root = self._preprocessed_root()
        cands = sorted(d for d in os.listdir(root) if d.startswith("Task%03d" % task_id) and os.path.isdir(os.path.join(root, d)))
        if len(cands) != 1:
            raise RuntimeError("expected exactly one preprocessed folder for task id %d under %s, found %s" % (task_id, root, cands))
        return os.path.join(root, cands[0])

    def _build_custom_splits(self, keys):
        """Folds 0-4: the five-fold splits of the individual datasets re-used (Task046 follows Task017's split for the shared
        images and deals its own cases out with a seeded shuffle); folds 5-11: leave-one-dataset-out (train == val)."""
        import os
        import pickle
        load = lambda f: pickle.load(open(f, 'rb'))
        fivefold = [{'train': [], 'val': []} for _ in range(5)]
        for task_id in np.unique([int(k.split("_")[0]) for k in keys]):
            if task_id != 46:
                per_task = load(os.path.join(self._task_folder(task_id), 'splits_final.pkl'))
                for f in range(5):
                    for part in ('train', 'val'):
                        fivefold[f][part] += ["%03d_" % task_id + c for c in per_task[f][part]]
            else:
                own = [k for k in keys if k.startswith("046_PAN")]
                np.random.RandomState(1234).shuffle(own)
                t17 = load(os.path.join(self._task_folder(17), 'splits_final.pkl'))
                for f in range(5):
                    for part in ('train', 'val'):
                        fivefold[f][part] += ["046_" + c for c in t17[f][part]]
                    val = own[f::5]
                    fivefold[f]['train'] += [k for k in own if k not in val]
                    fivefold[f]['val'] += val
        left_out = [("003_",), ("017_", "046_img"), ("064_",), ("010_",), ("007_",), ("055_",), ("008_",)]     # folds 5 .. 11
        custom = []
        for prefixes in left_out:
            rest = [k for k in keys if not any(k.startswith(p) for p in prefixes)]
            custom.append({'train': rest, 'val': rest})
        return fivefold + custom

    def do_split(self):
        """splits_custom.pkl (created by rank 0 when missing, the other ranks wait for the file); cases of a split that are not in
        the preprocessed folder are skipped with a warning."""
        import os
        import pickle
        import time
        from collections import OrderedDict
        keys = list(self.dataset.keys())
        if self.fold == "all":
            tr_keys = val_keys = keys
        else:
            f = os.path.join(self.dataset_directory, "splits_custom.pkl")
            if not os.path.isfile(f) and self.local_rank == 0:
                splits = self._build_custom_splits(keys)
                with open(f + ".tmp", 'wb') as h:
                    pickle.dump(splits, h)
                os.replace(f + ".tmp", f)            # atomic: another rank never unpickles a half-written file
            while not os.path.isfile(f):
                time.sleep(0.01)
            with open(f, 'rb') as h:
                splits = pickle.load(h)
            tr_keys, val_keys = list(splits[self.fold]['train']), list(splits[self.fold]['val'])
        tr_keys.sort()
        val_keys.sort()
        self.dataset_tr, self.dataset_val = OrderedDict(), OrderedDict()
        for dst, ks in ((self.dataset_tr, tr_keys), (self.dataset_val, val_keys)):
            for k in ks:
                if k in self.dataset:
                    dst[k] = self.dataset[k]
                else:
                    self.print_to_log_file('Warning %s is not in preprocessed folder (might be intentional)' % k)

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
            # the output of the SAME forward pass (reference :355-357), not a second one
            self.run_online_evaluation([self.train_step.last_logits], largs[0], largs[1])
        return l.detach().cpu().numpy(), ce.detach().cpu().numpy(), dc.detach().cpu().numpy()

    def run_online_evaluation(self, output, target, valid_regions, data=None):
        """Reference :372-410: hard (sigmoid > 0.5) tp/fp/fn per valid region on the full-resolution output — one counting kernel
        (exact integers) instead of the per-(sample, region) loop —, gathered over the ranks [W, B, C]; the lists keep the sum over
        the RANK axis only ([B, C] per iteration), exactly like the reference."""
        from .....distributed_utils import gather_over_ranks
        from .....loss_functions.fused_losses import _as_ndhwc, _target_flat
        from ...... import ops
        with torch.no_grad():
            if output is None:
                x = self.network.engine().forward(data, need_grad=False, all_heads=True)[0]               # NDHWC
            else:
                x = _as_ndhwc(output[0])
            valid, lut = self.train_step.loss_fn._masks(valid_regions, x.device)
            a = ops.Act(x)
            st = torch.empty((a.N, a.C, 3), dtype=torch.float32, device=x.device)
            ops.multitalent_hard_stats(a, _target_flat(target[0]), valid, lut, st)
            st = gather_over_ranks(st).cpu().numpy()                                                       # [W, B, C, 3]
        tp_hard, fp_hard, fn_hard = st[..., 0], st[..., 1], st[..., 2]
        self.online_eval_foreground_dc.append(list((2 * tp_hard) / (2 * tp_hard + fp_hard + fn_hard + 1e-8)))
        self.online_eval_tp.append(list(tp_hard.sum(0)))
        self.online_eval_fp.append(list(fp_hard.sum(0)))
        self.online_eval_fn.append(list(fn_hard.sum(0)))

    def finish_online_evaluation(self):
        """Reference :412-431: per (sample slot, region) Dice of the epoch's summed counts, mean over all of them."""
        tp, fp, fn = np.sum(self.online_eval_tp, 0), np.sum(self.online_eval_fp, 0), np.sum(self.online_eval_fn, 0)
        global_dc_per_class = [l for l in [2 * i / (np.clip(2 * i + j + k, a_min=1e-8, a_max=None)) for i, j, k in zip(tp, fp, fn)]
                               if not np.isnan(l).any()]
        self.all_val_eval_metrics.append(np.mean(global_dc_per_class))
        self.print_to_log_file("Average global foreground Dice:", str(global_dc_per_class))
        self.print_to_log_file("(interpret this as an estimate for the Dice of the different classes. This is not exact.)")
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []

    def _log_epoch_losses(self, tr, va):
        """the reference logs loss, BCE and Dice of the epoch (:702-724); run_iteration returns the three."""
        tr, va = np.asarray(tr, dtype=np.float64), np.asarray(va, dtype=np.float64)
        self.all_tr_losses.append(float(tr[:, 0].mean()))
        self.print_to_log_file("train loss : %.4f  ce: %.4f  dice: %.4f" % tuple(tr.mean(0)))
        self.all_val_losses.append(float(va[:, 0].mean()))
        self.print_to_log_file("validation loss: %.4f  ce: %.4f  dice: %.4f" % tuple(va.mean(0)))

    # ---- validation on the held-out cases (reference :129-322) ----------------------------------------------------------
    def validate(self, do_mirroring=True, use_sliding_window=True, step_size=0.5, save_softmax=True, use_gaussian=True,
                 overwrite=True, validation_folder_name='validation_raw', debug=False, all_in_gpu=False,
                 segmentation_export_kwargs=None, run_postprocessing_on_folds=False):
        """Cases strided over the ranks (`all_keys[local_rank::world]`); per case the sliding window, then from the SAME
        device-resident probabilities (a) one binary mask per region -> `<folder>_individual/<case>__<region>.nii.gz` and (b) the
        dataset's own label map (its valid regions painted in `MultiTalent_regions_class_order`) -> `<folder>/<case>.nii.gz`;
        rank 0 writes `summary_<dataset>.json`.  `run_postprocessing_on_folds` is ignored like in the reference."""
        import os
        import pickle
        from ......dataset_conversion.Task100_MultiTalent import MultiTalent_regions_class_order
        from ......evaluation.evaluator import aggregate_scores
        from ......inference.segmentation_export import save_segmentation_nifti_from_softmax
        current_mode = self.network.training
        self.network.eval()
        args = {'do_mirroring': do_mirroring, 'use_sliding_window': use_sliding_window, 'step_size': step_size,
                'save_softmax': save_softmax, 'use_gaussian': use_gaussian, 'overwrite': overwrite,
                'validation_folder_name': validation_folder_name, 'debug': debug, 'all_in_gpu': all_in_gpu,
                'segmentation_export_kwargs': segmentation_export_kwargs}
        out, mirror_axes = self._open_validation(do_mirroring, validation_folder_name, args)
        out_individual = os.path.join(self.output_folder, validation_folder_name + '_individual')
        os.makedirs(out_individual, exist_ok=True)
        force_separate_z, order, order_z = self._export_params(segmentation_export_kwargs)
        rank, world = self._validation_world()
        all_keys = list(self.dataset_val.keys())
        my_keys = all_keys[rank::world]
        pred_gt_tuples, valid_labels = {}, {}
        for k in all_keys:                 # every rank walks ALL keys: rank 0 needs the file pairs of every case (reference :198-201)
            with open(self.dataset[k]['properties_file'], 'rb') as f:
                properties = pickle.load(f)
            names = [i for i in MultiTalent_valid_regions.keys() if i.startswith("Task%03.0d_" % int(k.split('_')[0]))]
            assert len(names) == 1
            dataset_name = names[0]
            valid_labels.setdefault(dataset_name, properties['valid_labels'])
            fname = properties['list_of_data_files'][0].split("/")[-1][:-12]
            pred_gt_tuples.setdefault(dataset_name, []).append([os.path.join(out, fname + ".nii.gz"),
                                                                os.path.join(self.gt_niftis_folder or '', fname + ".nii.gz")])
            needed = overwrite or not os.path.isfile(os.path.join(out, fname + ".nii.gz")) or \
                (save_softmax and not os.path.isfile(os.path.join(out, fname + ".npz"))) or \
                any(not os.path.isfile(os.path.join(out_individual, fname + '__' + r + ".nii.gz")) for r in MultiTalent_regions)
            if not (needed and k in my_keys):
                continue
            probs = self._predict_validation_case(k, do_mirroring, mirror_axes, use_sliding_window, step_size, use_gaussian,
                                                  all_in_gpu)
            for r in MultiTalent_regions.keys():
                ch = MultiTalent_region_output_idx_mapping[r]
                save_segmentation_nifti_from_softmax(probs[ch:ch + 1], os.path.join(out_individual, fname + '__' + r + ".nii.gz"),
                                                     properties, order, ((1,),), None, None, None, None, force_separate_z,
                                                     order_z, verbose=False)
            idx = [MultiTalent_region_output_idx_mapping[i] for i in MultiTalent_valid_regions[dataset_name]]
            save_segmentation_nifti_from_softmax(probs[idx], os.path.join(out, fname + ".nii.gz"), properties, order,
                                                 MultiTalent_regions_class_order[dataset_name], None, None,
                                                 os.path.join(out, fname + ".npz") if save_softmax else None, None,
                                                 force_separate_z, order_z, verbose=False)
        self._validation_barrier()
        self.print_to_log_file("finished prediction")
        if rank == 0 and self.gt_niftis_folder is not None and os.path.isdir(self.gt_niftis_folder):
            self.print_to_log_file("evaluation of raw predictions")
            task = (self.dataset_directory or "").split("/")[-1]
            for dataset in pred_gt_tuples:
                aggregate_scores(pred_gt_tuples[dataset], labels=valid_labels[dataset],
                                 json_output_file=os.path.join(out, "summary_%s.json" % dataset),
                                 json_name=self.experiment_name + " val tiled %s" % str(use_sliding_window), json_author="Fabian",
                                 json_task=task)
        self.network.train(current_mode)
        self._validation_barrier()


class MultiTalent_trainer_ddp_2000ep(MultiTalent_trainer_ddp):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.max_num_epochs = 2000                                            # reference :796-808


class nnUNetTrainerV2_MultiTalent(MultiTalent_trainer_ddp):
    """Name used by BASELINE.json's north_star; behaviour of MultiTalent_trainer_ddp."""
