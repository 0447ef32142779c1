#!/usr/bin/env python3
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_bf16_gpu import _report
name, B, nbf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rep, eng = _report(name, B, nbf)
