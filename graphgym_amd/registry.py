"""Layer registry with the contract of graphgym/register.py:6-10,32-34.

If GraphGym itself is importable, its own ``layer_dict`` object is used — our layers land in the
very dictionary ``GeneralLayer`` resolves from (graphgym/models/layer.py:24,238).  Otherwise an
equivalent local dictionary is kept.  A duplicate key raises ``KeyError`` exactly like the reference.
"""
try:
    import graphgym.register as _gg_register
    layer_dict = _gg_register.layer_dict
    HAVE_GRAPHGYM = True
except Exception:
    _gg_register = None
    layer_dict = {}
    HAVE_GRAPHGYM = False

_DUPLICATE = 'Key {} is already pre-defined.'


def register(key, module, module_dict):
    """insert once; a second registration of ``key`` is an error, never a silent overwrite"""
    if key in module_dict:
        raise KeyError(_DUPLICATE.format(key))
    module_dict[key] = module


def register_layer(key, module):
    register(key, module, layer_dict)
