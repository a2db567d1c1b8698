"""htm_hashjoin_amd -- MI355X-native hash-join build+probe engine.

Python is only a thin host layer over the C ABI of libhtmjoin_hip.so
(include/htm_hashjoin.h): the operators below mirror the free functions of the
reference (anilshanbhag/HTM-HashJoin) by name and argument meaning.
"""
from ._lib import (  # noqa: F401
    HJ_OK, HJ_ERR_INVALID, HJ_ERR_NO_DEVICE, HJ_ERR_HIP, HJ_ERR_OOM, HJ_ERR_KEY_RANGE,
    HJ_ERR_UNKNOWN_ALGO, HJ_ERR_STATE, LIB_PATH, hj_params, hj_result, lib,
)
from .engine import (  # noqa: F401
    HashJoinError, HashJoinContext, NoCCHashBuild, AtomicHashBuild, HTMHashBuild, PRO,
    generate_data, generate_relation, device_count, SHARD_ONE_BASED, BUCKET_DTYPE,
)
