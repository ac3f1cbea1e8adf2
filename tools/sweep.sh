P=tools/queue_probe.py
python $P --pipes 4 --modes injected,dnn
python $P --pipes 4 --paf-shared-stream 1 --modes injected,dnn
python $P --pipes 4 --paf-shared-stream 2 --modes injected,dnn
python $P --pipes 6 --paf-shared-stream 1 --modes injected
