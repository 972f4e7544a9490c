import json, sys
d = json.load(open(sys.argv[1]))
print('%.1f M entries/s  %.2f ms/step  roofline %s %.3f' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']))
print(d['kernel_census_ms_per_step'])
