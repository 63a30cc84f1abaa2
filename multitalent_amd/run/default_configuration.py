"""`get_default_configuration` (reference run/default_configuration.py:33-83): where the plans, the preprocessed data and the
output of a (network, task, trainer, plans identifier) combination live.  Folders come from the same environment variables as
the reference's paths.py: `nnUNet_preprocessed`, `RESULTS_FOLDER` (+ `/nnUNet`)."""
import os
import pickle

from ..training.model_restore import find_trainer_class

default_plans_identifier = "nnUNetPlansv2.1"


def preprocessing_output_dir():
    d = os.environ.get('nnUNet_preprocessed')
    if d is None:
        raise RuntimeError("nnUNet_preprocessed is not defined: cannot locate plans and preprocessed data")
    return d


def network_training_output_dir():
    d = os.environ.get('RESULTS_FOLDER')
    if d is None:
        raise RuntimeError("RESULTS_FOLDER is not defined: cannot locate the training output folder")
    return os.path.join(d, "nnUNet")


def convert_id_to_task_name(task_id):
    """utilities/task_name_id_conversion.py, restricted to the preprocessed root."""
    start = "Task%03.0d" % int(task_id)
    cands = sorted(d for d in os.listdir(preprocessing_output_dir()) if d.startswith(start))
    if len(cands) != 1:
        raise RuntimeError("expected exactly one preprocessed task folder starting with %s, found %s" % (start, cands))
    return cands[0]


def get_default_configuration(network, task, network_trainer, plans_identifier=default_plans_identifier):
    assert network in ['2d', '3d_lowres', '3d_fullres', '3d_cascade_fullres'], \
        "network can only be one of the following: '2d', '3d_lowres', '3d_fullres', '3d_cascade_fullres'"
    if network != '3d_fullres':
        raise NotImplementedError("only 3d_fullres is on the MultiTalent path")
    dataset_directory = os.path.join(preprocessing_output_dir(), task)
    plans_file = os.path.join(dataset_directory, plans_identifier + "_plans_3D.pkl")
    with open(plans_file, 'rb') as f:
        plans = pickle.load(f)
    stages = list(plans['plans_per_stage'].keys())
    stage = stages[-1]
    trainer_class = find_trainer_class(network_trainer)
    output_folder_name = os.path.join(network_training_output_dir(), network, task, network_trainer + "__" + plans_identifier)
    batch_dice = len(stages) > 1
    return plans_file, output_folder_name, dataset_directory, batch_dice, stage, trainer_class
