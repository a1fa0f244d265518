"""Import-compatibility shim: `import pylibwholegraph.torch as wgth` resolves to the MI355X-native
`wholegraph_amd.torch` (embedding gather / scatter / gradient-apply path only — see DESIGN.md §6 for what
is out of scope)."""
import sys

import wholegraph_amd
import wholegraph_amd.torch as _torch_layer

__version__ = wholegraph_amd.__version__
sys.modules[__name__ + ".torch"] = _torch_layer
torch = _torch_layer
