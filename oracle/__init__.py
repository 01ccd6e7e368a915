"""oracle/ — CPU restatement of the reference's algorithms for the filter -> Aggregated / Mutations path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import or execute anything in this directory; the product (lapis-silo_amd/) never does.
"""
