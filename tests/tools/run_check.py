#!/usr/bin/env python
"""Developer aid (GPU box): runs one check of tests/engine_checks.py against a given build of the library.
usage: python tests/tools/run_check.py <lib.so|default> <check name> [python literal kwargs]"""
import ast
import os
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: F401,E402
import engine_checks as ec  # noqa: E402
lib = None if sys.argv[1] == 'default' else sys.argv[1]
kw = ast.literal_eval(sys.argv[3]) if len(sys.argv) > 3 else {}
try:
    print(sys.argv[1], sys.argv[2], 'ok ->', getattr(ec, sys.argv[2])(lib, **kw))
except AssertionError as ex:
    print(sys.argv[1], sys.argv[2], 'FAILED:', str(ex)[:300])
