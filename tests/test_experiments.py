"""experiments/: patches of experiments that were measured and not kept.  They are evidence, not product: this only checks that each
still applies to the commit its README names (`git apply --check` on an export of that commit; nothing is built or run)."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_4_patches_are_tracked_described_and_apply_to_their_base(tmp_path):
    readme = open(os.path.join(ROOT, "experiments", "r04", "README.md")).read()
    patches = sorted(glob.glob(os.path.join(ROOT, "experiments", "r04", "*.patch")))
    assert len(patches) == 5
    for p in patches:
        assert "`%s`" % os.path.basename(p) in readme, "experiments/r04/README.md does not describe %s" % os.path.basename(p)
    assert "Memory access fault" in readme and "0xfc" in readme           # the round's GPU fault and its cause are written down
    base = re.search(r"applies to commit `([0-9a-f]{7,40})`", readme).group(1)
    if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", base + "^{commit}"], capture_output=True).returncode != 0:
        pytest.skip("no git history here (a snapshot without .git)")
    tar = subprocess.run(["git", "-C", ROOT, "archive", base, "haskell-path-tracer_amd", "include", "tools", "tests"], capture_output=True, check=True).stdout
    subprocess.run(["tar", "-x", "-C", str(tmp_path)], input=tar, check=True)
    for p in patches:
        res = subprocess.run(["git", "apply", "--check", p], cwd=tmp_path, capture_output=True, text=True)
        assert res.returncode == 0, "%s does not apply to %s:\n%s" % (os.path.basename(p), base, res.stderr)


def test_round_6_patches_are_tracked_described_and_apply_to_their_base(tmp_path):
    readme = open(os.path.join(ROOT, "experiments", "r06", "README.md")).read()
    patches = sorted(glob.glob(os.path.join(ROOT, "experiments", "r06", "*.patch")))
    assert len(patches) == 2
    for p in patches:
        assert "`%s`" % os.path.basename(p) in readme, "experiments/r06/README.md does not describe %s" % os.path.basename(p)
    assert "garbage collector" in readme                                  # what the copy kernel's suspicion turned out to be
    base = re.search(r"applies to commit `([0-9a-f]{7,40})`", readme).group(1)
    if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", base + "^{commit}"], capture_output=True).returncode != 0:
        pytest.skip("no git history here (a snapshot without .git)")
    tar = subprocess.run(["git", "-C", ROOT, "archive", base, "haskell-path-tracer_amd"], capture_output=True, check=True).stdout
    subprocess.run(["tar", "-x", "-C", str(tmp_path)], input=tar, check=True)
    for p in patches:
        res = subprocess.run(["git", "apply", "--check", p], cwd=tmp_path, capture_output=True, text=True)
        assert res.returncode == 0, "%s does not apply to %s:\n%s" % (os.path.basename(p), base, res.stderr)
