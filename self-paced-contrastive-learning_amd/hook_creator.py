"""Mirror of the reference's ``hook_creator.py`` (repo root): config sections -> TrainerHooks.

``InfonceParams`` -> ``create_infonce_hooks`` (:16-18), ``SPInfonceParams`` -> ``create_sp_infonce_hooks`` with the
trainer's ``max_epoch`` (:19-23); ``DiscreteMIConsistencyParams`` belongs to a comparison baseline outside the hot path
 and is refused here -- during pre-training the reference refuses it too (:24-26)."""
from .semi_seg.hooks import create_infonce_hooks, create_sp_infonce_hooks


def create_hook_from_config(model, config, is_pretrain=False):
    data_name = config["Data"]["name"]
    max_epoch = config["Trainer"]["max_epoch"]
    hooks = []
    if "InfonceParams" in config:
        hooks.append(create_infonce_hooks(model=model, data_name=data_name, **config["InfonceParams"]))
    if "SPInfonceParams" in config:
        hooks.append(create_sp_infonce_hooks(model=model, data_name=data_name, max_epoch=max_epoch,
                                             **config["SPInfonceParams"]))
    if "DiscreteMIConsistencyParams" in config:
        if is_pretrain:
            raise RuntimeError("DiscreteMIConsistencyParams are not supported for pretrain stage")
        raise NotImplementedError("DiscreteMIConsistencyParams: comparison baseline outside the HIP hot path")
    return hooks
