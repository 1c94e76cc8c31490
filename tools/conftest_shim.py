"""Reads the JSON metadata stored next to the golden vectors (same helper as tests/conftest.py::golden_json)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden_meta(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as f:
        return json.loads(bytes(f["meta"]).decode())
