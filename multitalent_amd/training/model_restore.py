"""Plugin discovery and checkpoint restore (reference training/model_restore.py:23-148): trainers are found by class
NAME by walking the network_training package; a checkpoint is `<name>.model` + `<name>.model.pkl`."""
import importlib
import os
import pickle
import pkgutil

import torch

import multitalent_amd


def recursive_find_python_class(folder, trainer_name, current_module):
    tr = None
    for _, modname, ispkg in pkgutil.iter_modules(folder):
        if not ispkg:
            m = importlib.import_module(current_module + "." + modname)
            if hasattr(m, trainer_name):
                return getattr(m, trainer_name)
    for _, modname, ispkg in pkgutil.iter_modules(folder):
        if ispkg:
            tr = recursive_find_python_class([os.path.join(folder[0], modname)], trainer_name, current_module + "." + modname)
            if tr is not None:
                return tr
    return tr


def find_trainer_class(name):
    folder = [os.path.join(multitalent_amd.__path__[0], "training", "network_training")]
    tr = recursive_find_python_class(folder, name, "multitalent_amd.training.network_training")
    if tr is None:
        raise RuntimeError("Could not find the trainer class %s in multitalent_amd.training.network_training" % name)
    return tr


def restore_model(pkl_file, checkpoint=None, train=False, fp16=None):
    """reference :44-100."""
    with open(pkl_file, 'rb') as f:
        info = pickle.load(f)
    init, name = info['init'], info['name']
    tr = find_trainer_class(name)
    trainer = tr(*init)
    if fp16 is not None:
        trainer.fp16 = fp16
    trainer.process_plans(info['plans'])
    trainer.plans = info['plans']
    if checkpoint is not None:
        trainer.load_checkpoint(checkpoint, train)
    return trainer


def load_model_and_checkpoint_files(folder, folds=None, mixed_precision=None, checkpoint_name="model_best"):
    """reference :109-148: returns (trainer, list of checkpoint dicts)."""
    if isinstance(folds, str):
        folds = [os.path.join(folder, "all")]
        assert os.path.isdir(folds[0]), "no output folder for fold %s found" % folds
    elif isinstance(folds, (list, tuple)):
        if len(folds) == 1 and folds[0] == "all":
            folds = [os.path.join(folder, "all")]
        else:
            folds = [os.path.join(folder, "fold_%d" % i) for i in folds]
        assert all(os.path.isdir(i) for i in folds), "list of folds specified but not all output folders are present"
    elif isinstance(folds, int):
        folds = [os.path.join(folder, "fold_%d" % folds)]
        assert os.path.isdir(folds[0]), "output folder missing for fold %s" % folds
    elif folds is None:
        folds = sorted(os.path.join(folder, d) for d in os.listdir(folder) if d.startswith("fold") and os.path.isdir(os.path.join(folder, d)))
    else:
        raise ValueError("Unknown value for folds. Type: %s. Expected: list of int, int, str or None" % str(type(folds)))
    trainer = restore_model(os.path.join(folds[0], "%s.model.pkl" % checkpoint_name), fp16=mixed_precision)
    trainer.output_folder = folder
    trainer.output_folder_base = folder
    trainer.update_fold(0)
    trainer.initialize(False)
    files = [os.path.join(i, "%s.model" % checkpoint_name) for i in folds]
    params = [torch.load(i, map_location='cpu', weights_only=False) for i in files]
    return trainer, params
