"""Boundary check against the reference's real registry module, where it is present.

graphgym/register.py imports in this container; the hot-path modules do not.  Run in a
subprocess with the reference on PYTHONPATH so that graphgym_amd.registry binds to the reference's own
``layer_dict`` — the object GeneralLayer resolves layer_type keys from (graphgym/models/layer.py:24,238).
Skipped on the GPU box, where /root/reference does not exist."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import graphgym.register as ref                      # the reference's module, unmodified
import graphgym_amd.registry as ours
import graphgym_amd.graphgym_plugin as plugin
out = {}
out["same_dict_object"] = ours.layer_dict is ref.layer_dict
out["keys_in_reference_dict"] = sorted(k for k in plugin.ALL_KEYS if ref.layer_dict.get(k) is plugin.ALL_KEYS[k])
try:
    ref.register_layer("gcnidconv", object)          # the reference's own duplicate check now sees our entry
    out["dup"] = None
except KeyError as e:
    out["dup"] = str(e)
try:
    ours.register("x", 1, {"x": 0})
except KeyError as e:
    out["ours_msg"] = str(e)
try:
    ref.register("x", 1, {"x": 0})
except KeyError as e:
    out["ref_msg"] = str(e)
layer = ref.layer_dict["Tfg-idgcn"](8, 16, bias=True)
out["params"] = sorted(n for n, _ in layer.named_parameters())
print(json.dumps(out))
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "graphgym")), reason="reference checkout not present")
def test_plugin_lands_in_the_reference_registry():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, REF]))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["same_dict_object"]
    assert len(out["keys_in_reference_dict"]) == 18
    assert out["dup"] is not None and "already pre-defined" in out["dup"]
    assert out["ours_msg"] == out["ref_msg"]          # identical KeyError text (register.py:8)
    assert out["params"] == ["model.bias", "model.kernel", "model.kernel_id"]
