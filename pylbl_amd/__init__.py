"""MI355X-native line-by-line molecular-lines engine (drop-in for pyLBL's lines backend)."""
