"""`DictConfig`: the smallest object with the `config` duck type the trainers use (the reference's
`parse_config.ConfigParser`, parse_config.py:15-141, provides the same members and works unchanged): item access
into the JSON dict, `initialize(name, module, *args, **kw)` reflection, `get_logger`, `save_dir`, `resume`.  Used by
tests and bench on machines without the reference checkout; it performs NO command-line parsing and creates no directories
except `save_dir`."""
from __future__ import annotations

import inspect
import json
import logging
from pathlib import Path


class DictConfig:
    def __init__(self, config, save_dir=None, resume=None):
        if isinstance(config, (str, Path)):
            with open(config) as f:
                config = json.load(f)
        self._config = config
        self.resume = Path(resume) if resume is not None else None
        self._save_dir = Path(save_dir if save_dir is not None else config['trainer']['save_dir'])
        self._save_dir.mkdir(parents=True, exist_ok=True)
        self.log_levels = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}

    def initialize(self, name, module, *args, index=None, **kwargs):
        """parse_config.py:88-113."""
        if index is None:
            module_name = self[name]['type']
            module_args = dict(self[name]['args'])
            assert all(k not in module_args for k in kwargs), 'Overwriting kwargs given in config file is not allowed'
            module_args.update(kwargs)
        else:
            module_name = self[name][index]['type']
            module_args = dict(self[name][index]['args'])
        signature = inspect.signature(getattr(module, module_name).__init__)
        for param in signature.parameters.keys():
            if param not in module_args and param in self.config:
                module_args[param] = self[param]
        return getattr(module, module_name)(*args, **module_args)

    def __getitem__(self, name):
        return self._config[name]

    def get_logger(self, name, verbosity=2):
        assert verbosity in self.log_levels
        logger = logging.getLogger(name)
        logger.setLevel(self.log_levels[verbosity])
        return logger

    @property
    def config(self):
        return self._config

    @property
    def save_dir(self):
        return self._save_dir
