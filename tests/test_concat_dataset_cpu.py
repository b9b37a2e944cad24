"""ConcatDataset mirror against the reference class (tests/golden/concat_dataset.json from tools/gen_golden.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools", "ref_shims"))     # tiny_ds.Tiny


def test_concat_dataset_matches_reference():
    from fsnet_amd.vision_base.data.datasets.dataset_utils import ConcatDataset
    rec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "concat_dataset.json")))
    ds = ConcatDataset(rec["cfgs"], **rec["common"])
    assert len(ds) == rec["length"] == 15
    assert [{k: (int(v) if k == "value" else v) for k, v in ds[i].items()} for i in range(len(ds))] == rec["items"]
    assert ds._determine_index(5) == (1, 0) and ds._determine_index(7) == (1, 2) and ds._determine_index(8) == (2, 0)
