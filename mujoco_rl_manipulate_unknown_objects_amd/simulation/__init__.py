"""Environment package; mirrors the reference's ``simulation`` package layout."""
