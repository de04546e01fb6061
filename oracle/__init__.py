"""CPU oracle for the bake_shading hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``iris_amd``) never does.  See ``oracle/iris_oracle.c`` for the restatement
and its pinning status.
"""
from .oracle import *  # noqa: F401,F403
