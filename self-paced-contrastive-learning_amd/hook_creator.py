"""Mirror of the reference's repo-root ``hook_creator.py``: which sections of the merged config turn into TrainerHooks.

Contract (hook_creator.py:10-28): ``InfonceParams`` builds plain InfoNCE hooks, ``SPInfonceParams`` the self-paced ones
(they also get the trainer's ``max_epoch`` for their age-parameter schedule); both receive ``Data.name``.
``DiscreteMIConsistencyParams`` is a comparison baseline outside the hot path: during pre-training the
reference raises RuntimeError for it and so does this mirror; otherwise it is refused as not implemented."""
from .semi_seg import hooks as _hooks

# config section -> (factory in semi_seg.hooks, does the factory take max_epoch?)
_SECTIONS = (
    ("InfonceParams", "create_infonce_hooks", False),
    ("SPInfonceParams", "create_sp_infonce_hooks", True),
)
_BASELINE_SECTION = "DiscreteMIConsistencyParams"


def create_hook_from_config(model, config, is_pretrain=False):
    common = {"model": model, "data_name": config["Data"]["name"]}
    built = []
    for section, factory, wants_epochs in _SECTIONS:
        params = config.get(section)
        if params is None:
            continue
        extra = {"max_epoch": config["Trainer"]["max_epoch"]} if wants_epochs else {}
        built.append(getattr(_hooks, factory)(**common, **extra, **params))
    if _BASELINE_SECTION in config:
        if is_pretrain:
            raise RuntimeError(f"{_BASELINE_SECTION} are not supported for pretrain stage")
        raise NotImplementedError(f"{_BASELINE_SECTION}: comparison baseline outside the HIP hot path")
    return built
