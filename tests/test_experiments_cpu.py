"""tools/experiments/*.patch are measured-and-not-adopted variants kept out of the product tree (DESIGN.md sections 4.1 / 4c name their
numbers); each must still apply to the sources it was cut from, or the A/B scripts under tools/calls/ that need it cannot be re-run."""
import glob, os, shutil, subprocess
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "tools", "experiments", "*.patch")))


@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_experiment_patch_applies(patch):
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    r = subprocess.run(["patch", "--dry-run", "-p1", "-s", "-i", patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_experiment_is_named_in_design():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for p in PATCHES:
        assert "tools/experiments/" + os.path.basename(p) in design, p
