"""`python ahv_run.py <script>` / `python -m 3dahv_amd <script>`: an UNCHANGED reference script finds the hot path's callables
already rebound when its own `from utils import rotate_volume` runs (3dahv_amd/__main__.py).  A stand-in checkout here (the
reference's utils.py needs cv2 / matplotlib, which this image lacks); the kernels themselves are the GPU suite's business."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_checkout(root):
    (root / "modules").mkdir()
    (root / "utils.py").write_text("def rotate_volume(volume, rotation_matrix, padding_mode='zeros'):\n    return 'reference'\n")
    (root / "modules" / "__init__.py").write_text("")
    (root / "modules" / "modules.py").write_text(
        "class Feature_Aligner:\n"
        "    def forward_3d2d(self, x):\n        return 'reference'\n"
        "    def forward_2d3d(self, a, b, random_mask=True, mask_ratio=0.25):\n        return 'reference'\n")
    (root / "test_script.py").write_text(
        "import sys\n"
        "from utils import rotate_volume\n"
        "from modules.modules import Feature_Aligner\n"
        "import importlib\n"
        "patch = importlib.import_module('3dahv_amd.patch')\n"
        "assert __name__ == '__main__'\n"
        "print('ARGV', sys.argv[1:])\n"
        "print('PATCHED', rotate_volume is patch._hip_rotate_volume, Feature_Aligner.forward_3d2d is patch._hip_forward_3d2d,\n"
        "      Feature_Aligner.forward_2d3d is patch._hip_forward_2d3d, hasattr(Feature_Aligner, 'verify_hypotheses'), patch._defer)\n"
        "sys.exit(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0)\n")


def run(cmd, cwd, env=None):
    e = {k: v for k, v in os.environ.items() if k != "AHV_PATCH_DEFER"}
    e.update(env or {})
    return subprocess.run(cmd, cwd=cwd, env=e, capture_output=True, text=True, timeout=300)


def test_runner_rebinds_before_the_script_imports(tmp_path):
    make_checkout(tmp_path)
    out = run([sys.executable, os.path.join(REPO, "ahv_run.py"), "test_script.py", "--config", "config.yaml"], tmp_path)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ARGV ['--config', 'config.yaml']" in out.stdout and "PATCHED True True True True True" in out.stdout
    # the module form, the script's exit code, and the switch that keeps every line its own kernel
    out = run([sys.executable, "-m", "3dahv_amd", "--ahv-no-defer", "test_script.py", "3"], tmp_path, {"PYTHONPATH": REPO})
    assert out.returncode == 3 and "PATCHED True True True True False" in out.stdout, (out.stdout, out.stderr[-2000:])
    # usage errors: no script, a missing script, an unknown option
    for args in ([], ["nope.py"], ["--ahv-what", "test_script.py"]):
        bad = run([sys.executable, os.path.join(REPO, "ahv_run.py")] + args, tmp_path)
        assert bad.returncode == 2 and bad.stderr.strip(), args
