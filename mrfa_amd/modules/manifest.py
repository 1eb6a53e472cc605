"""state_dict manifests ([name, shape, dtype]) of the product modules, for checkpoint-compatibility checks."""
from __future__ import annotations


def manifest_of(module):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()]


def raft_flow_manifest(cfg):
    from .raft import RaftFlow
    return manifest_of(RaftFlow(**cfg))
