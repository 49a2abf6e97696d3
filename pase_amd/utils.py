"""Mirror of the parts of pase/utils.py the PASE(+) training path uses."""
import json

import torch.nn as nn

from .losses import ContextualizedLoss


def worker_parser(cfg_fname, batch_acum=1, device="cpu", do_losses=True, frontend=None):
    """JSON -> {"regr": [...], "cls": [...]}; every "loss": "<nn name>" string becomes
    ContextualizedLoss(getattr(nn, name)(), r) (pase/utils.py:53-68).  GAN losses (:70-88) are not
    part of any shipped PASE(+) worker cfg."""
    with open(cfg_fname, "r") as cfg_f:
        cfg_list = json.load(cfg_f)
    if do_losses:
        for _type, cfg_all in cfg_list.items():
            for i, cfg in enumerate(cfg_all):
                loss_name = cfg_all[i]["loss"]
                if hasattr(nn, loss_name):
                    r_frames = cfg_all[i].get("r", None)
                    cfg_all[i]["loss"] = ContextualizedLoss(getattr(nn, loss_name)(), r=r_frames)
                else:
                    raise NotImplementedError("pase_amd worker_parser: loss %r" % loss_name)
    return cfg_list


def strip_transforms(minions_cfg):
    """train.py:64 pops the 'transform' sub-dict of each worker cfg before the model is built."""
    for _type, cfg_all in minions_cfg.items():
        for cfg in cfg_all:
            cfg.pop("transform", None)
    return minions_cfg
