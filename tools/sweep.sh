P=tools/queue_probe.py
python $P --pipes 4 --modes injected,dnn
python $P --pipes 4 --paf-own-stream --modes injected,dnn
python $P --pipes 6 --paf-own-stream --modes injected
python $P --pipes 3 --paf-own-stream --modes injected
python $P --pipes 2 --paf-own-stream --modes injected
