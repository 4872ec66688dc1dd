"""The repo's model files against the reference's own model patches.

The reference holds `mjpc/tasks/*/*.xml.patch` files that turn upstream models into the ones its tasks load; every `+` and context line of
such a patch is a line of the patched model. tools/make_model_provenance.py hashed the simulated elements on those lines (tag +
attributes; rendering-only elements and attributes left out) into tests/golden/model_provenance.json. Here: every one of them is an element
of the repo's restatement of that model -- joint ranges, geom sizes, friction, actuator gains, contact pairs and sensors are the
reference's numbers, not retyped approximations. When /root/reference is present the fixture itself is checked against the patches."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_model_provenance as mp  # noqa: E402

FIXTURE = json.load(open(os.path.join(ROOT, "tests", "golden", "model_provenance.json")))


@pytest.mark.parametrize("name", sorted(mp.MODELS))
def test_every_patched_element_is_in_the_restated_model(name):
    _patch, mine = mp.MODELS[name]
    have = {mp.digest(e) for e in mp.file_elements(os.path.join(ROOT, mine))}
    want = FIXTURE[name]["elements"]
    assert len(want) == FIXTURE[name]["count"] and len(want) >= 10
    missing = [h for h in want if h not in have]
    assert not missing, f"{len(missing)} of {len(want)} elements of {FIXTURE[name]['patch']} are not in {mine}"


def test_the_humanoid_patch_rewrites_the_whole_model():
    """humanoid.xml.patch is one hunk over the whole upstream file, so the patched model is fully determined by it: the repo's file holds
    no simulated element beyond the patch's (the other models keep upstream lines the patches do not show)"""
    _patch, mine = mp.MODELS["humanoid"]
    have = {mp.digest(e) for e in mp.file_elements(os.path.join(ROOT, mine))}
    extra = have - set(FIXTURE["humanoid"]["elements"])
    assert not extra, f"{len(extra)} elements of {mine} are not in the patch"


@pytest.mark.skipif(not os.path.isdir(mp.REF), reason="the reference tree is not on this machine")
def test_the_fixture_is_what_the_patches_say():
    assert mp.build() == FIXTURE
