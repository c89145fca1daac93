"""Config base class: nested classes are the config tree (reference mqe/envs/base/base_config.py:34-54)."""
import inspect


class BaseConfig:
    def __init__(self):
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        """Replace every nested class attribute by an instance of it, recursively."""
        for name in dir(obj):
            if name == "__class__":
                continue
            member = getattr(obj, name)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, name, inst)
                BaseConfig.init_member_classes(inst)
