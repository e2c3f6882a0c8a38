import sys, os, subprocess
print("priority range", __import__("torch").cuda.Stream.priority_range() if hasattr(__import__("torch").cuda.Stream, "priority_range") else None)
