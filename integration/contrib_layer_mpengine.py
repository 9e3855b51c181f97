"""Copy (or symlink) this file into  graphgym/contrib/layer/  of a GraphGym checkout.

graphgym/contrib/layer/__init__.py:1-4 globs every *.py of that directory into __all__ and
graphgym/models/layer.py:11 star-imports it, so this module runs before layer_dict is resolved
(layer.py:238).  It must sort after idconv.py (it does: 'm' > 'i') so that the reference's own
registrations exist when ours replace them.
"""
import graphgym_amd.graphgym_plugin  # noqa: F401  (registers / overrides the layer_type keys)
