"""Stand-in package so `from Bio import SeqIO` in the reference resolves (tools/make_golden.py)."""
