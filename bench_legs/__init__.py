"""The legs of bench.py, one module per concern; bench.py itself holds only main(): the timed region and the JSON line."""
