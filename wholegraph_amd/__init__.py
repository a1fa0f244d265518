"""wholegraph_amd — MI355X-native WholeMemory embedding gather / scatter / gradient-apply path.

Drop-in for that one path of rapidsai/wholegraph: the ``wholememory_*`` C ABI lives in
``libwholegraph.so`` (sources under ``wholegraph_amd/csrc``, headers under ``include/wholememory``) and
``wholegraph_amd.torch`` mirrors ``pylibwholegraph.torch`` (comm / tensor / embedding / ops).
"""
__version__ = "0.1.0"
