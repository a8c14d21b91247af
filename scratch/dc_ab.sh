python -m pytest tests/test_hip_parity_r2.py -q -m gpu -k "dc_rows or multicoil or 640" 2>&1 | tail -5
bash scratch/dc_prof.sh "${1:-w368}" "1 15 640 368"
bash scratch/dc_prof.sh "${1:-w368}" "8 1 640 368"
bash scratch/dc_prof.sh "${1:-w368}" "2 15 640 368"
