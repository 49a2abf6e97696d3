"""pase_amd -- MI355X-native PASE / PASE+ self-supervised training step.

The public surface mirrors the reference package (santi-pdp/pase):
    pase_amd.frontend.wf_builder / WaveFe        <- pase.models.frontend
    pase_amd.pase.pase                           <- pase.models.pase
    pase_amd.minions, pase_amd.losses, pase_amd.utils.worker_parser
    pase_amd.trainer.trainer / LR_Scheduler      <- pase.models.WorkerScheduler
`install_as_pase()` registers those modules under the reference's import paths so existing scripts
(`from pase.models.frontend import wf_builder`) pick up the HIP implementation unchanged.
"""
import sys
import types

__all__ = ["install_as_pase"]


def install_as_pase():
    """Alias this package as `pase` (only the modules of the accelerated path)."""
    from . import frontend, losses, minions, modules, pase as pase_mod, trainer, utils

    def mod(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    root = mod("pase")
    models = mod("pase.models")
    ws = mod("pase.models.WorkerScheduler")
    mn = mod("pase.models.Minions")
    for name, target in (("pase.models.frontend", frontend), ("pase.models.modules", modules),
                         ("pase.models.pase", pase_mod), ("pase.losses", losses), ("pase.utils", utils),
                         ("pase.models.Minions.minions", minions), ("pase.models.Minions.cls_minions", minions),
                         ("pase.models.WorkerScheduler.trainer", trainer),
                         ("pase.models.WorkerScheduler.lr_scheduler", trainer)):
        sys.modules[name] = target
    root.models, root.losses, root.utils = models, losses, utils
    models.frontend, models.modules, models.pase = frontend, modules, pase_mod
    models.WorkerScheduler, models.Minions = ws, mn
    ws.trainer, ws.lr_scheduler = trainer, trainer
    mn.minions, mn.cls_minions = minions, minions
    return root
