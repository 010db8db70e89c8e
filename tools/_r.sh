set -e
mkdir -p gpurun_out/r5z
python -m pytest tests/test_lstm_trainer.py -m gpu -q -x 2>&1 | tail -1
python tools/bench_lstm_trainer.py > gpurun_out/r5z/bench_lstm2.json 2> gpurun_out/r5z/bench_lstm.err
cat gpurun_out/r5z/bench_lstm2.json
