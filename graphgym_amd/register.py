"""Layer registry with the semantics of graphgym/register.py:6-10,32-34.

If GraphGym itself is importable its own ``layer_dict`` is used (so our layers land in
the very registry ``GeneralLayer`` resolves from, graphgym/models/layer.py:24,238);
otherwise an identical local registry is kept.  ``register`` raises ``KeyError`` on a
duplicate key exactly like the reference.
"""
try:
    import graphgym.register as _gg_register
    layer_dict = _gg_register.layer_dict
    HAVE_GRAPHGYM = True
except Exception:
    _gg_register = None
    layer_dict = {}
    HAVE_GRAPHGYM = False


def register(key, module, module_dict):
    if key in module_dict:
        raise KeyError('Key {} is already pre-defined.'.format(key))
    module_dict[key] = module


def register_layer(key, module):
    register(key, module, layer_dict)
