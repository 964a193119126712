"""CPU oracle for the dots.ocr hot path — TEST INFRASTRUCTURE, never shipped, never on the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/model.py (vision tower + LM + greedy generate) and oracle/image_processor.py for what
is restated from where and which parts are pinned / unpinned.
"""
