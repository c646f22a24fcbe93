"""Stand-in `diffusers` for tests/test_reference_pin.py (see tests/refstub/README.md).  TEST INFRASTRUCTURE ONLY."""
__version__ = "0.24.0+refstub"
