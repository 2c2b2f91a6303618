"""Helper of tests/test_host_cpu.py::test_install_dropin_is_order_independent (run as a script: a fresh interpreter per order).
argv[1]: path_first | install_first | no_reference; argv[2]: the reference's tree.  Third-party packages that the reference's
UNREPLACED scripts import and this image lacks (torchvision, tifffile, tensorboard) get inert stand-ins; everything else is real:
with the drop-in installed, the reference's dataset.py, cmdiad_runner.py, main.py and hallucination_network_pretrain.py must import,
its own utils.misc must stay its own, and the redirected names must resolve to this package."""
import importlib
import os
import sys
import types
REF = sys.argv[2] if len(sys.argv) > 2 else "/root/reference"
sys.dont_write_bytecode=True
order=sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def stub(name, **attrs):
    m=types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name]=m; return m
tv=stub('torchvision'); tr=stub('torchvision.transforms'); tv.transforms=tr
class _T:
    def __init__(self,*a,**k): pass
    def __call__(self,x): return x
for n in ('Compose','Resize','ToTensor','Normalize'): setattr(tr,n,_T)
tr.InterpolationMode=types.SimpleNamespace(BICUBIC=3,NEAREST=0)
v2=stub('torchvision.transforms.v2'); tr.v2=v2
for n in ('Compose','Resize','ToTensor','Normalize','ToImage','ToDtype'): setattr(v2,n,_T)
stub('tifffile', imread=lambda p: None)
import torch.utils
tb=stub('torch.utils.tensorboard', SummaryWriter=_T); torch.utils.tensorboard=tb
import cmdiad_amd
if order=='path_first':
    sys.path.insert(0,REF); cmdiad_amd.install_dropin()
elif order=='install_first':
    cmdiad_amd.install_dropin(); sys.path.insert(0,REF)
else:
    cmdiad_amd.install_dropin()
if order!='no_reference':
    for name in ('dataset','cmdiad_runner','main','hallucination_network_pretrain'):
        m=importlib.import_module(name)
    import utils.misc
    assert os.path.realpath(utils.misc.__file__).startswith(os.path.realpath(REF))
    import dataset
    assert dataset.resize_organized_pc.__module__.startswith('cmdiad_amd')
from feature_extractors import multiple_features
import utils.lr_sched as l
from utils import lr_sched as l2
from models.hrnet import HRNet
assert multiple_features.__name__.startswith('cmdiad_amd') and l is l2 and HRNet.__module__.startswith('cmdiad_amd')
print(order,'ok')
