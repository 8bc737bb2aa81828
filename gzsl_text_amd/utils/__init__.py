"""Callers either side of the train step: batch staging and eval scoring, mirroring the reference's
`utils/helper.py` and `utils/metrics.py` names."""
