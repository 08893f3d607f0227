"""CPU oracle (test infrastructure only; never imported by voicesplit_amd/)."""
