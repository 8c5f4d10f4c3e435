import sys; sys.path.insert(0,'.')
import torch, numpy as np
from viterbidecodercpp_amd import *
from tests.helpers import make_table_config
for cid in range(8):
    code = COMMON_CODES[cid]
    pc, table, config = make_table_config(code, "SOFT16")
    d = BatchDecoder(table, config)
    print(code.name, 'plan', d.plan, 'G', list(d._handle.info.polynomials[:code.R]), 'linear', d._handle.info.table_is_linear)
