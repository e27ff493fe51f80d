"""CPU oracle package -- test infrastructure, never imported by egovlp_amd (see egovlp_oracle.py)."""
