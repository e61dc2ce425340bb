"""The example driver keeps the reference driver's command line (overiva_oneshot.py:72-116)."""
import os
import re

from conftest import REPO


def test_flags_match_reference_driver():
    src = open(os.path.join(REPO, "examples", "overiva_oneshot.py")).read()
    for flag, default in (('"-a", "--algo"', None), ('"-d", "--dist"', None), ('"-i", "--init"', None),
                          ('"-m", "--mics"', "default=5"), ('"-s", "--srcs"', "default=2"),
                          ('"-n", "--n_iter"', "default=51"), ('"--no_cb"', None)):
        assert flag in src
        if default:
            line = [l for l in src.splitlines() if flag in l][0]
            assert default in line
    assert re.search(r'algo_choices = \["auxiva", "auxiva_pca", "overiva"\]', src)
    assert "framesize = 4096" in src
