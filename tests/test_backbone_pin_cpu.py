"""Known-answer checks for the MobileNetV2 backbone that do NOT pass through this repository's own
builder or stand-in: the published torchvision==0.9 `mobilenet_v2` state-dict layout and parameter counts, written out
here as literals.  torchvision is a third-party dependency absent from /root/reference and from this image, so the
sub-graph's VALUES stay "parity unpinned"  until a torchvision-produced tensor is available; what these
tests pin is everything the reference's own code relies on at that boundary (park2019.py:107-108,130-132; revgrad.py:71):
key names, shapes, counts, and -- when SPB_MOBILENETV2_WEIGHTS points at a real `mobilenet_v2-*.pth` -- strict loading.
"""
import os

import pytest
import torch

# torchvision 0.9 mobilenet_v2: features[k] for k = 1..17 as (expand ratio, in, out, stride) -- literal table of the
# published architecture (Sandler et al. 2018, table 2, width 1.0), not derived from this repository's config lists
BLOCKS = {
    1: (1, 32, 16, 1),
    2: (6, 16, 24, 2), 3: (6, 24, 24, 1),
    4: (6, 24, 32, 2), 5: (6, 32, 32, 1), 6: (6, 32, 32, 1),
    7: (6, 32, 64, 2), 8: (6, 64, 64, 1), 9: (6, 64, 64, 1), 10: (6, 64, 64, 1),
    11: (6, 64, 96, 1), 12: (6, 96, 96, 1), 13: (6, 96, 96, 1),
    14: (6, 96, 160, 2), 15: (6, 160, 160, 1), 16: (6, 160, 160, 1),
    17: (6, 160, 320, 1),
}
# published totals: mobilenet_v2 3 504 872 parameters; classifier Linear(1280, 1000) 1 281 000; features[18]
# (1x1 320->1280 + BN) 412 160  =>  features[:-1] 1 811 712
TV_TOTAL, TV_CLASSIFIER, TV_LAST = 3504872, 1281000, 412160


def bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def published_feature_keys():
    keys = [("features.0.0.weight", (32, 3, 3, 3))] + bn_keys("features.0.1", 32)
    for k, (t, cin, cout, s) in BLOCKS.items():
        p = "features.%d.conv." % k
        hid = cin * t
        i = 0
        if t != 1:
            keys += [(p + "0.0.weight", (hid, cin, 1, 1))] + bn_keys(p + "0.1", hid)
            i = 1
        keys += [(p + "%d.0.weight" % i, (hid, 1, 3, 3))] + bn_keys(p + "%d.1" % i, hid)
        keys += [(p + "%d.weight" % (i + 1), (cout, hid, 1, 1))] + bn_keys(p + "%d" % (i + 2), cout)
    return keys


def test_published_counts():
    keys = published_feature_keys()
    n = sum(int(torch.Size(s).numel()) for k, s in keys if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n == TV_TOTAL - TV_CLASSIFIER - TV_LAST == 1811712
    assert len(keys) == 1 + 5 + (2 + 10) + 16 * (3 + 15)                  # 306 tensors: 51 convs + 51 BatchNorms x 5


def test_hip_plan_and_module_surface_use_the_published_layout():
    """the C++ plan's arena (libspb_hip.so, no GPU needed to query it), the nn.Module surface and the oracle all expose
    exactly the published keys with `features.` -> `base.` (park2019.py:108 wraps features[:-1] in self.base)"""
    from speedplusbaseline_amd.engine import KrnEngine
    from speedplusbaseline_amd.nets.park2019 import KeypointRegressionNet
    from oracle import krn_oracle as O
    want = [("base." + k[len("features."):], s) for k, s in published_feature_keys()]
    eng = KrnEngine(11)
    plan = {n: s for n, s, _, _ in eng.param_infos}
    plan.update({n: s for n, s, _, _ in eng.buffer_infos})
    plan.update({n: () for n in eng.bn_names})
    model = KeypointRegressionNet(11)
    msd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    osd = O.krn_param_shapes(11)
    for k, s in want:
        assert plan[k] == s, k
        assert msd[k] == s, k
        assert tuple(osd[k]) == s, k
    base_keys = [k for k in msd if k.startswith("base.")]
    assert len(base_keys) == len(want) and base_keys == [k for k, _ in want]          # same order as torchvision's state_dict
    # what the reference's own code asserts about the backbone: 18 modules, tap depth 96 at base[13], 320 channels out
    assert len(model.base) == 18
    assert msd["base.13.conv.2.weight"][0] == 96 and msd["base.17.conv.2.weight"][0] == 320
    assert sum(v.numel() for k, v in model.state_dict().items() if k.startswith("base.") and "running" not in k and "num_batches" not in k) == 1811712
    assert sum(p.numel() for p in model.parameters()) == 5643862


def test_backbone_loader_takes_a_torchvision_state_dict(tmp_path, monkeypatch):
    """park2019.py:107 `mobilenet_v2(pretrained=True)`: the loader accepts torchvision's file (all of `features.*`,
    `classifier.*`), drops features[18] and the classifier, and loads the rest strictly"""
    from speedplusbaseline_amd.nets.park2019 import KeypointRegressionNet
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, s in published_feature_keys():
        sd[k] = torch.zeros(s, dtype=torch.int64) if k.endswith("num_batches_tracked") else torch.rand(s, generator=g)
    sd["features.18.0.weight"] = torch.rand(1280, 320, 1, 1, generator=g)
    for k, s in bn_keys("features.18.1", 1280):
        sd[k] = torch.zeros(s, dtype=torch.int64) if k.endswith("num_batches_tracked") else torch.rand(s, generator=g)
    sd["classifier.1.weight"] = torch.rand(1000, 1280, generator=g); sd["classifier.1.bias"] = torch.rand(1000, generator=g)
    path = tmp_path / "mobilenet_v2-b0353104.pth"
    torch.save(sd, path)
    monkeypatch.setenv("SPB_MOBILENETV2_WEIGHTS", str(path))
    model = KeypointRegressionNet(11)
    got = model.state_dict()
    for k, _ in published_feature_keys():
        assert torch.equal(got["base." + k[len("features."):]], sd[k]), k
    sd.pop("features.3.conv.1.1.running_var")                              # an incomplete file must not load silently
    torch.save(sd, path)
    with pytest.raises(RuntimeError):
        KeypointRegressionNet(11)


@pytest.mark.skipif(not os.path.isfile(os.environ.get("SPB_MOBILENETV2_WEIGHTS", "")), reason="no torchvision mobilenet_v2-*.pth on this box")
def test_real_torchvision_checkpoint_loads_strictly():
    from speedplusbaseline_amd.nets.park2019 import KeypointRegressionNet
    sd = torch.load(os.environ["SPB_MOBILENETV2_WEIGHTS"], map_location="cpu")
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == TV_TOTAL
    model = KeypointRegressionNet(11)
    got = model.state_dict()
    for k, s in published_feature_keys():
        assert tuple(sd[k].shape) == s and torch.equal(got["base." + k[len("features."):]], sd[k]), k
