"""Regenerate tests/golden fixtures (run in the build container where /root/reference exists).
Data only: the .dat example file and numbers quoted from svo.wout / SURVEY.md Appendix B."""
import gzip
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/aps_example"

with open(os.path.join(REF, "svo_hr.dat"), "rb") as src, gzip.GzipFile(os.path.join(HERE, "svo_hr.dat.gz"), "wb", 9, mtime=0) as dst:
    shutil.copyfileobj(src, dst)

meta = {
    "source": "aps_example/svo.wout:89-99 and SURVEY.md Appendix B",
    "a_angstrom": 3.858560,
    "b_inv_angstrom": 1.628376,
    "num_wann": 3,
    "nrpts": 1331,
    "eig_known": {
        "G": [[0.0, 0.0, 0.0], [11.447075, 11.447075, 11.447075]],
        "X": [[0.5, 0.0, 0.0], [11.562485, 13.320933, 13.320933]],
        "M": [[0.5, 0.5, 0.0], [13.301791, 13.301791, 13.659419]],
        "R": [[0.5, 0.5, 0.5], [13.876733, 13.876733, 13.876733]],
        "k123": [[0.1, 0.2, 0.3], [12.351303, 12.824980, 12.909626]],
    },
    "nn_diag_R100": [-0.255871, -0.026000, -0.255871],
    "max_abs_re": 12.975161,
}
with open(os.path.join(HERE, "svo_meta.json"), "w") as fh:
    json.dump(meta, fh, indent=1)
