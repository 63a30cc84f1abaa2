"""`predict_MultiTalent` driver on the HIP engine (reference nnunet/inference/predict_MultiTalent.py:39-108,127-266,269-376):
same entry points (`predict_cases`, `predict_from_folder`, `check_input_folder_and_return_caseIDs`, `main`), same file naming
(`<out>/individual/<case>_<region>.nii.gz`, one binary mask per MultiTalent region), same fold ensembling.

What is different underneath: the reference preprocesses in background processes and exports in a process pool because both are
CPU-bound (seconds to minutes per case); here `trainer.preprocess_patient` resamples on the device (milliseconds), the sliding
window keeps the probabilities in HBM, and every region's mask is resampled + thresholded by `mt_resample_classify` from that one
resident tensor — `num_threads_preprocessing` / `num_threads_nifti_save` are accepted and ignored."""
import argparse
import os
import pickle
import shutil

import numpy as np
import torch

from ..dataset_conversion.Task100_MultiTalent import MultiTalent_region_output_idx_mapping, MultiTalent_regions
from ..training.model_restore import load_model_and_checkpoint_files
from .segmentation_export import save_segmentation_nifti_from_softmax


def _export_params(trainer, segmentation_export_kwargs):
    if segmentation_export_kwargs is None:
        p = trainer.plans.get('segmentation_export_params')
        if p is not None:
            return p['force_separate_z'], p['interpolation_order'], p['interpolation_order_z']
        return None, 1, 0
    k = segmentation_export_kwargs
    return k['force_separate_z'], k['interpolation_order'], k['interpolation_order_z']


def predict_cases(model, list_of_lists, output_filenames, folds, save_npz, num_threads_preprocessing=None,
                  num_threads_nifti_save=None, segs_from_prev_stage=None, do_tta=True, mixed_precision=True,
                  overwrite_existing=False, all_in_gpu=False, step_size=0.5, checkpoint_name="model_final_checkpoint",
                  segmentation_export_kwargs=None, disable_postprocessing=False):
    """reference :127-266.  list_of_lists: [[case0_0000.nii.gz, ...], ...]; output_filenames: [case0.nii.gz, ...]."""
    assert len(list_of_lists) == len(output_filenames)
    if segs_from_prev_stage is not None:
        raise NotImplementedError("cascade inputs (segs_from_prev_stage) are not on the MultiTalent path")
    cleaned = []
    for o in output_filenames:
        dr, f = os.path.split(o)
        if len(dr) > 0:
            os.makedirs(dr, exist_ok=True)
        if not f.endswith(".nii.gz"):
            f = os.path.splitext(f)[0] + ".nii.gz"
        cleaned.append(os.path.join(dr, f))
    if not overwrite_existing:
        print("number of cases:", len(list_of_lists))
        todo = [i for i, j in enumerate(cleaned) if (not os.path.isfile(j)) or (save_npz and not os.path.isfile(j[:-7] + '.npz'))]
        cleaned = [cleaned[i] for i in todo]
        list_of_lists = [list_of_lists[i] for i in todo]
        print("number of cases that still need to be predicted:", len(cleaned))
    print("loading parameters for folds,", folds)
    trainer, params = load_model_and_checkpoint_files(model, folds, mixed_precision=mixed_precision, checkpoint_name=checkpoint_name)
    force_separate_z, order, order_z = _export_params(trainer, segmentation_export_kwargs)
    print("starting prediction...")
    for input_files, output_filename in zip(list_of_lists, cleaned):
        try:
            d, _, dct = trainer.preprocess_patient(input_files, return_device=True)
        except KeyboardInterrupt:
            raise
        except Exception as e:                        # the reference's workers skip a broken case and report it (:79-83)
            print("error in", input_files)
            print(e)
            continue
        print("predicting", output_filename)
        dr, f = os.path.split(output_filename)
        os.makedirs(os.path.join(dr, 'individual'), exist_ok=True)
        probs = None
        for p in params:
            trainer.load_checkpoint_ram(p, False)
            cur = trainer.predict_preprocessed_data_return_seg_and_softmax(
                d, do_mirroring=do_tta, mirror_axes=trainer.data_aug_params['mirror_axes'], use_sliding_window=True,
                step_size=step_size, use_gaussian=True, all_in_gpu=all_in_gpu, mixed_precision=mixed_precision,
                return_device_tensors=True)[1]
            # `cur` aliases the network's sliding-window cache: an ensemble accumulates into its own copy
            probs = (cur.clone() if len(params) > 1 else cur) if probs is None else probs.add_(cur)
        if len(params) > 1:
            probs /= len(params)
        tf = trainer.plans.get('transpose_forward')
        if tf is not None:
            tb = trainer.plans.get('transpose_backward')
            if list(tb) != [0, 1, 2]:
                probs = probs.permute(0, *[int(i) + 1 for i in tb]).contiguous()
        for region in MultiTalent_regions.keys():
            ch = MultiTalent_region_output_idx_mapping[region]
            stem = os.path.join(dr, 'individual', f[:-7] + '_' + region)
            save_segmentation_nifti_from_softmax(probs[ch:ch + 1], stem + '.nii.gz', dct, order, ((1,),), None, None,
                                                 stem + '.npz' if save_npz else None, None, force_separate_z, order_z, verbose=False)
        print("inference done.")


def check_input_folder_and_return_caseIDs(input_folder, expected_num_modalities):
    """reference :269-303: <case>_0000.nii.gz ... per modality; missing files raise."""
    print("This model expects %d input modalities for each image" % expected_num_modalities)
    files = sorted(i for i in os.listdir(input_folder) if i.endswith(".nii.gz") and os.path.isfile(os.path.join(input_folder, i)))
    assert len(files) > 0, "input folder did not contain any images (expected to find .nii.gz file endings)"
    case_ids = np.unique([i[:-12] for i in files])
    remaining, missing = list(files), []
    for c in case_ids:
        for n in range(expected_num_modalities):
            expected = c + "_%04.0d.nii.gz" % n
            if not os.path.isfile(os.path.join(input_folder, expected)):
                missing.append(expected)
            else:
                remaining.remove(expected)
    print("Found %d unique case ids, here are some examples:" % len(case_ids), list(case_ids[:10]))
    if len(remaining) > 0:
        print("found %d unexpected remaining files in the folder. Here are some examples:" % len(remaining), remaining[:10])
    if len(missing) > 0:
        print("Some files are missing:")
        print(missing)
        raise RuntimeError("missing files in input_folder")
    return case_ids


def predict_from_folder(model, input_folder, output_folder, folds, save_npz, num_threads_preprocessing=None,
                        num_threads_nifti_save=None, lowres_segmentations=None, part_id=0, num_parts=1, tta=True,
                        mixed_precision=True, overwrite_existing=True, mode='normal', overwrite_all_in_gpu=None, step_size=0.5,
                        checkpoint_name="model_final_checkpoint", segmentation_export_kwargs=None, disable_postprocessing=True):
    """reference :306-376: standard naming -> predict_cases; cases are strided over `num_parts` processes."""
    os.makedirs(output_folder, exist_ok=True)
    assert os.path.isfile(os.path.join(model, "plans.pkl")), "Folder with saved model weights must contain a plans.pkl file"
    shutil.copy(os.path.join(model, 'plans.pkl'), output_folder)
    with open(os.path.join(model, "plans.pkl"), 'rb') as f:
        expected_num_modalities = pickle.load(f)['num_modalities']
    case_ids = check_input_folder_and_return_caseIDs(input_folder, expected_num_modalities)
    output_files = [os.path.join(output_folder, i + ".nii.gz") for i in case_ids]
    all_files = sorted(i for i in os.listdir(input_folder) if i.endswith(".nii.gz"))
    list_of_lists = [[os.path.join(input_folder, i) for i in all_files if i[:len(j)].startswith(j) and len(i) == (len(j) + 12)]
                     for j in case_ids]
    if lowres_segmentations is not None:
        raise NotImplementedError("cascade inputs (lowres_segmentations) are not on the MultiTalent path")
    if mode != "normal":
        raise ValueError("unrecognized mode. Must be normal. Fast or fastest not implemented")
    all_in_gpu = False if overwrite_all_in_gpu is None else overwrite_all_in_gpu
    return predict_cases(model, list_of_lists[part_id::num_parts], output_files[part_id::num_parts], folds, save_npz,
                         num_threads_preprocessing, num_threads_nifti_save, None, tta, mixed_precision=mixed_precision,
                         overwrite_existing=overwrite_existing, all_in_gpu=all_in_gpu, step_size=step_size,
                         checkpoint_name=checkpoint_name, segmentation_export_kwargs=segmentation_export_kwargs,
                         disable_postprocessing=disable_postprocessing)


def main(argv=None):
    """reference :379-540: the same flags; one process per GPU, `--part_id/--num_parts` (or RANK/WORLD_SIZE under a launcher)
    stride the cases over the processes (no communication, like the reference)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", '--input_folder', required=True)
    ap.add_argument('-o', "--output_folder", required=True)
    ap.add_argument('-m', '--model_output_folder', required=True)
    ap.add_argument('-f', '--folds', nargs='+', default='None')
    ap.add_argument('-z', '--save_npz', action='store_true')
    ap.add_argument('-l', '--lowres_segmentations', default='None')
    ap.add_argument("--part_id", type=int, default=None)
    ap.add_argument("--num_parts", type=int, default=None)
    ap.add_argument("--local_rank", default=None, type=int)
    ap.add_argument("--local-rank", dest='local_rank', type=int)
    ap.add_argument("--num_threads_preprocessing", default=6, type=int)
    ap.add_argument("--num_threads_nifti_save", default=2, type=int)
    ap.add_argument("--tta", type=int, default=1)
    ap.add_argument("--overwrite_existing", type=int, default=1)
    ap.add_argument("--mode", type=str, default="normal")
    ap.add_argument("--all_in_gpu", type=str, default="None")
    ap.add_argument("--step_size", type=float, default=0.5)
    ap.add_argument('--disable_mixed_precision', default=False, action='store_true')
    ap.add_argument('-chk', '--checkpoint_name', default='model_final_checkpoint')
    a = ap.parse_args(argv)
    local_rank = a.local_rank if a.local_rank is not None else int(os.environ.get('LOCAL_RANK', 0))
    part_id = a.part_id if a.part_id is not None else int(os.environ.get('RANK', 0))
    num_parts = a.num_parts if a.num_parts is not None else int(os.environ.get('WORLD_SIZE', 1))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    folds = a.folds
    if isinstance(folds, list):
        if not (folds[0] == 'all' and len(folds) == 1):
            folds = [int(i) for i in folds]
    elif folds == "None":
        folds = None
    else:
        raise ValueError("Unexpected value for argument folds")
    if a.tta not in (0, 1):
        raise ValueError("Unexpected value for tta, Use 1 or 0")
    if a.overwrite_existing not in (0, 1):
        raise ValueError("Unexpected value for overwrite, Use 1 or 0")
    assert a.all_in_gpu in ['None', 'False', 'True']
    predict_from_folder(a.model_output_folder, a.input_folder, a.output_folder, folds, a.save_npz, a.num_threads_preprocessing,
                        a.num_threads_nifti_save, None if a.lowres_segmentations == "None" else a.lowres_segmentations, part_id,
                        num_parts, bool(a.tta), mixed_precision=not a.disable_mixed_precision,
                        overwrite_existing=bool(a.overwrite_existing), mode=a.mode,
                        overwrite_all_in_gpu={'None': None, 'True': True, 'False': False}[a.all_in_gpu], step_size=a.step_size,
                        checkpoint_name=a.checkpoint_name)


if __name__ == "__main__":
    main()
