"""Drop-in registration of the engine's layers under GraphGym's ``layer_type`` keys.

Importing this module is the whole integration:

  * the ID-GNN keys 'idconv', 'gcnidconv', 'sageidconv', 'gatidconv', 'ginidconv'
    (registered by graphgym/contrib/layer/idconv.py:444-448 in the reference),
  * the TF-family keys 'Tfg-gcnconv', 'Tfg-sageconv', 'Tfg-gatconv', 'Tfg-ginconv',
    'Tfg-idgcn', 'Tfg-idsage', 'Tfg-idgat', 'Tfg-idgin' (config/*_tf/*.yaml:29, dispatched
    by main_zd.py:299-308 in the reference),
  * the built-in keys 'gcnconv', 'sageconv', 'gatconv', 'ginconv', 'generalconv'
    (graphgym/models/layer.py:224-235).

``register_layer`` raises KeyError on a duplicate (register.py:6-10), and built-ins shadow
registered keys (layer.py:238), so taking over an existing key is done by assignment into the
dictionaries — ``install(override=True)``, the default.
"""
from . import layers as L
from . import registry as R

ID_KEYS = {
    'idconv': L.GeneralIDConv,
    'gcnidconv': L.GCNIDConv,
    'sageidconv': L.SAGEIDConv,
    'gatidconv': L.GATIDConv,
    'ginidconv': L.GINIDConv,
}
TF_KEYS = {
    'Tfg-gcnconv': L.TfgGCNConv,
    'Tfg-sageconv': L.TfgSAGEConv,
    'Tfg-gatconv': L.TfgGATConv,
    'Tfg-ginconv': L.TfgGINConv,
    'Tfg-idgcn': L.TfgIDGCN,
    'Tfg-idsage': L.TfgIDSAGE,
    'Tfg-idgat': L.TfgIDGAT,
    'Tfg-idgin': L.TfgIDGIN,
}
BUILTIN_KEYS = {
    'gcnconv': L.GCNConv,
    'sageconv': L.SAGEConv,
    'gatconv': L.GATConv,
    'ginconv': L.GINConv,
    'generalconv': L.GeneralConv,
}
ALL_KEYS = {**ID_KEYS, **TF_KEYS, **BUILTIN_KEYS}


def install(override=True):
    """Register every key; returns the list of keys now served by the engine.

    override=False keeps the reference's semantics strictly: keys that already exist are
    left alone (register_layer would raise KeyError)."""
    taken = []
    dicts = [R.layer_dict]
    try:  # the resolved dictionaries GeneralLayer actually indexes (layer.py:24,238)
        import sys
        for name in ("graphgym.models.layer", "graphgym.models.layer_pyg"):
            mod = sys.modules.get(name)
            if mod is not None and hasattr(mod, "layer_dict"):
                dicts.append(mod.layer_dict)
    except Exception:
        pass
    for key, cls in ALL_KEYS.items():
        for d in dicts:
            if key in d and d[key] is not cls:
                if override:
                    d[key] = cls
            elif key not in d:
                if d is R.layer_dict:
                    R.register_layer(key, cls)
                else:
                    d[key] = cls
        if R.layer_dict.get(key) is cls:
            taken.append(key)
    return taken


installed_keys = install(override=True)
