"""yaml -> attribute-style Config (same behaviour as reference config/load_config.py:6-43)."""

import yaml


class Config:
    @staticmethod
    def from_yaml(filepath):
        with open(filepath, "r") as fp:
            return Config.from_dict(yaml.safe_load(fp))

    @staticmethod
    def from_dict(data_map):
        config = Config()
        for name, value in data_map.items():
            if isinstance(value, dict):
                value = Config.from_dict(value)
            setattr(config, name, value)
        return config

    def to_dict(self):
        ret = {}
        for k, v in self.__dict__.items():
            ret[k] = v.to_dict() if isinstance(v, Config) else v
        return ret
