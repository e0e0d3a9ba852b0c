import json, sys
d = json.loads(sys.stdin.read())
c = d["config"]
print("hwq", c.get("hw_queues"), "P", c.get("passes_in_flight"), "S", c.get("streams_per_gpu"), "ms", d["ms_per_step"], "audio-s/s", d["value"], "roof", d["roofline"]["frac"])
