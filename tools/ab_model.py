"""Runs bench.py with vqa_amd/model_prev.py (a copy of an older model.py) in place of vqa_amd.model: same-box
A/B of host-side changes.  python tools/ab_model.py [bench arguments]"""
import importlib.util, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import vqa_amd._lib  # noqa: F401  (the package's own modules first)
spec = importlib.util.spec_from_file_location('vqa_amd.model', os.path.join(root, 'vqa_amd', 'model_prev.py'))
mod = importlib.util.module_from_spec(spec)
sys.modules['vqa_amd.model'] = mod
spec.loader.exec_module(mod)
sys.argv = [os.path.join(root, 'bench.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
