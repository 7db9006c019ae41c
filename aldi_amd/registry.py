"""Detectron2-style name -> object registry (the reference's plugin mechanism:
aldi/distill.py:17,33, aldi/align.py:11, detectron2 META_ARCH_REGISTRY used at aldi/model.py:14)."""


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}
        self.__doc__ = ""

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map
