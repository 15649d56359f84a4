"""The reference's `target:` / `params:` plugin surface, without omegaconf or ldm.

* instantiate_from_config / get_obj_from_str: same contract as [UPSTREAM] ldm/util.py, which the reference
  calls at train.py:445 (model), src/models/autoencoder.py:86,103,104 (loss, pose MLPs),
  train.py:452,463 and src/data/preprocessing/data_modules.py:83,89.
* Config: the OmegaConf subset train.py uses (load, merge, from_dotlist, attribute access, pop/get) on top
  of PyYAML (train.py:134-148).
* configure_learning_rate: train.py:356-392.
"""
import copy
import importlib

import yaml


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    try:
        return getattr(importlib.import_module(module, package=None), cls)
    except ModuleNotFoundError:
        # train.py:229 targets `pytorch_lightning.callbacks.ModelCheckpoint`; where pytorch_lightning is absent (this image) the
        # callbacks of that name built for trainer.Trainer stand in -- only for that one package prefix, everything else still raises
        if module == "pytorch_lightning.callbacks":
            from . import callbacks
            if hasattr(callbacks, cls):
                return getattr(callbacks, cls)
        raise


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__" or config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    if isinstance(params, Config):
        params = params.to_container()
    return get_obj_from_str(config["target"])(**params)


class Config(dict):
    """dict with attribute access, recursively; just enough of OmegaConf.DictConfig for the OD-VAE configs."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, (list, tuple)):
            return [Config._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def to_container(self):
        def unwrap(v):
            if isinstance(v, Config):
                return {k: unwrap(x) for k, x in v.items()}
            if isinstance(v, list):
                return [unwrap(x) for x in v]
            return v
        return unwrap(self)

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(self.to_container(), memo))

    # ---- OmegaConf-style constructors -----------------------------------------------------------------
    @staticmethod
    def create(data=None):
        return Config(data or {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return Config(yaml.safe_load(f) or {})

    @staticmethod
    def from_dotlist(items):
        """["a.b.c=1", "x=[1,2]"] -> nested config; values parsed as YAML scalars/lists like OmegaConf does."""
        cfg = Config()
        for item in items:
            key, _, raw = item.partition("=")
            node = cfg
            parts = key.lstrip("-").split(".")
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], Config):
                    node[p] = Config()
                node = node[p]
            node[parts[-1]] = yaml.safe_load(raw) if raw != "" else None
        return cfg

    @staticmethod
    def merge(*configs):
        """Later configs win; dicts merge recursively, everything else (lists included) is replaced."""
        out = Config()

        def rec(dst, src):
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    rec(dst[k], v)
                else:
                    dst[k] = copy.deepcopy(v.to_container() if isinstance(v, Config) else v)
        for c in configs:
            rec(out, c)
        return out


def configure_learning_rate(config, model, trainer_config, scale_lr=True, ngpu=None):
    """model.learning_rate = accumulate * ngpu * batch_size * base_lr (train.py:371-388).
    The reference reads ngpu from lightning trainer.devices when running on GPUs and uses 1 otherwise."""
    bs = config.data.params.batch_size
    base_lr = config.model.base_learning_rate
    if ngpu is None:
        ngpu = trainer_config.get("devices", 1) if trainer_config.get("accelerator", "cpu") == "gpu" else 1
        if not isinstance(ngpu, int):
            ngpu = 1
    accumulate = trainer_config.get("accumulate_grad_batches", 1)
    trainer_config["accumulate_grad_batches"] = accumulate
    model.learning_rate = accumulate * ngpu * bs * base_lr if scale_lr else base_lr
    return model
