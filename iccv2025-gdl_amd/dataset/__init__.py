"""Stand-ins for the reference dataset package (see _synthetic.py)."""
