"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's hot path).

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this package; the product (kg_instance_segmentation_amd/) never does.
Parity status: PINNED against tests/golden/*.npz generated from the reference by
tools/gen_goldens.py (see DESIGN.md section "Oracle").
"""
