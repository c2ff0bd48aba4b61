"""Hyper-parameter files: yaml -> a tree of attribute-style nodes (behaviour of reference
config/load_config.py:6-43: `Config.from_yaml`, `Config.from_dict`, `.to_dict()`, nested dicts become
nested Config objects, everything else -- lists included -- is kept as is)."""

import yaml


class Config:
    def __init__(self, **entries):
        for key, value in entries.items():
            setattr(self, key, Config(**value) if isinstance(value, dict) else value)

    @classmethod
    def from_dict(cls, mapping):
        return cls(**mapping)

    @classmethod
    def from_yaml(cls, filepath):
        with open(filepath, "r") as stream:
            return cls.from_dict(yaml.safe_load(stream))

    def to_dict(self):
        return {key: value.to_dict() if isinstance(value, Config) else value
                for key, value in vars(self).items()}

    def __repr__(self):
        return f"Config({self.to_dict()!r})"
