"""ORACLE -- test infrastructure only (see oracle/README in DESIGN.md section 3).

CPU restatement of the reference's Mip-NeRF 360 per-ray path.  Importable only
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package never imports it.
"""
