"""Alias package: the reference's import paths (`mmfn_utils.models.model_vec:MMFN`, used by
run_steps/config/train.yaml:12-15 entry points and by team_code/e2e_agent/mmfn_*.py:13) resolve to
the MI355X-native implementation in mmfn_amd."""
