"""The plugin contract between trainers / epochers and regularisation hooks, as the reference defines it in
``contrastyou/hooks/base.py`` (TrainerHook :23-35 with the unique-name check :11-20, CombineTrainerHook :38-49,
EpocherHook :52-86, CombineEpochHook :89-118) -- same class names, methods and behaviour, own implementation:

* a *trainer hook* is an ``nn.Module`` that owns the learnable parts of a regulariser (projector, ...) and, called
  once per epoch, manufactures the matching *epocher hook*;
* an *epocher hook* is a bundle of phase callbacks (``PHASES``) around the forward pass and the regularisation term,
  plus ``__call__(**kwargs) -> loss`` and ``close()``;
* the ``Combine*`` variants fan a call out to several hooks and add the losses up.
"""
import itertools
import weakref

from torch import nn

PHASES = ("before_forward_pass", "after_forward_pass", "before_regularization", "after_regularization")


class _HookNameRegistry(type):
    """A TrainerHook constructed with the literal keyword ``hook_name=...`` claims that name; claiming it twice is a
    ``ValueError`` (positional names and other keywords are not tracked -- exactly the reference's rule)."""
    _claimed = set()

    def __call__(cls, *args, **kwargs):
        if "hook_name" not in kwargs:
            return super().__call__(*args, **kwargs)
        name = kwargs["hook_name"]
        if name in _HookNameRegistry._claimed:
            raise ValueError(name)
        _HookNameRegistry._claimed.add(name)
        return super().__call__(*args, **kwargs)


class TrainerHook(nn.Module, metaclass=_HookNameRegistry):
    def __init__(self, hook_name: str):
        super().__init__()
        self._hook_name = hook_name

    @property
    def learnable_modules(self):
        """modules whose parameters the trainer's optimizer must see (none by default)"""
        return []

    def parameters(self, recurse: bool = True):
        return itertools.chain.from_iterable(m.parameters(recurse=recurse) for m in self.learnable_modules)


class CombineTrainerHook(TrainerHook):
    """several trainer hooks behind one; its epocher hook is the CombineEpochHook of theirs"""

    def __init__(self, *trainer_hook):
        super().__init__("")
        self._hooks = nn.ModuleList(trainer_hook)

    @property
    def learnable_modules(self):
        return self._hooks

    def __call__(self):
        return CombineEpochHook(*(hook() for hook in self._hooks))


class EpocherHook:
    """per-epoch side of a hook; ``set_epocher`` hands it weak proxies of the epocher and of its meter interface"""
    meters = None

    def __init__(self, name: str) -> None:
        self._name = name

    def set_epocher(self, epocher):
        self._epocher = weakref.proxy(epocher)
        self.meters = weakref.proxy(epocher.meters)
        self.configure_meters(self.meters)

    epocher = property(lambda self: self._epocher)

    def configure_meters(self, meters):
        return meters

    def __call__(self, **kwargs):
        return None

    def close(self):
        return None

    # ---- hipGraph replay of the epocher's step (stepgraph.py; no counterpart in the reference)
    def graph_key(self):
        """A hashable description of every HOST value this hook's ``__call__`` bakes into its kernel launches (weights,
        age parameter, shapes of its own), or None when a captured step cannot stand in for a call -- the default: a hook
        that does not say so is never replayed.  Per-step host data must reach the kernels through
        ``epocher.stage.bind`` (label vectors do)."""
        return None

    def after_replay(self):
        """called after each replayed step (the hook's ``__call__`` did not run): deferred host-side checks"""
        return None


def _silent_phase(self, **kwargs):
    return None


class CombineEpochHook(EpocherHook):
    def __init__(self, *epocher_hook: EpocherHook) -> None:
        self._epocher_hook = tuple(epocher_hook)
        # hooks that pool the SAME tapped feature to (1, 1) share one pooling pass (and one feature gradient): see
        # semi_seg/hooks/infonce.py `_two_views` (several meta-labels on one encoder pass)
        groups = {}
        for h in self._epocher_hook:
            key = getattr(h, "shared_pool_key", None)
            if key is not None:
                groups.setdefault(key, []).append(h)
        for members in groups.values():
            if len(members) > 1:
                for h in members:
                    h._share_pool = True
                # ... and, when their projector heads have one shape, run as ONE batched projection (one launch per
                # layer for all heads; `batchable_heads` decides)
                fn = getattr(members[0], "batchable_heads", None)
                if fn is not None and fn(members):
                    for h in members:
                        h._batch_group = members

    def _broadcast(self, method, *args, **kwargs):
        return [getattr(hook, method)(*args, **kwargs) for hook in self._epocher_hook]

    def set_epocher(self, epocher):
        self._broadcast("set_epocher", epocher)

    def __call__(self, **kwargs):
        losses = self._broadcast("__call__", **kwargs)
        total = losses[0] if losses else 0  # sum() would start from 0 + loss: one more launch for the same value
        for extra in losses[1:]:
            total = total + extra
        return total

    def close(self):
        self._broadcast("close")

    def graph_key(self):
        keys = tuple(h.graph_key() if hasattr(h, "graph_key") else None for h in self._epocher_hook)
        return None if any(k is None for k in keys) else ("combine",) + keys

    def after_replay(self):
        for h in self._epocher_hook:
            if hasattr(h, "after_replay"):
                h.after_replay()


def _fan_out(phase):
    def run(self, **kwargs):
        self._broadcast(phase, **kwargs)
    run.__name__ = phase
    return run


for _phase in PHASES:
    setattr(EpocherHook, _phase, _silent_phase)
    setattr(CombineEpochHook, _phase, _fan_out(_phase))
del _phase
