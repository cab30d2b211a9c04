/* Reads every capture file named on the command line through xe_pcap_* with
 * several buffer sizes; built with the address and undefined-behaviour
 * sanitizers by tests/test_cpu_pcap.py. */
#include <stdio.h>
#include <stdlib.h>
#include "../include/xe_gpu.h"
int main(int argc, char **argv)
{
	unsigned long total = 0;
	for (int a = 1; a < argc; a++) {
		static const size_t caps[] = { 70000, 1 << 20 };
		static const size_t maxd[] = { 1, 5, 4096 };
		for (int c = 0; c < 2; c++) for (int m = 0; m < 3; m++) for (int port = 0; port < 2; port++) {
			xe_pcap *p = NULL;
			int rc = xe_pcap_open(argv[a], &p);
			if (rc == 0) {
				uint8_t *data = malloc(caps[c]);
				uint64_t *offs = malloc((maxd[m] + 1) * 8);
				uint32_t *src = malloc(maxd[m] * 4);
				long n;
				while ((n = xe_pcap_read(p, port ? 2055 : 0, data, caps[c], offs, src, maxd[m])) > 0) {
					total += (unsigned long)n;
					if (offs[n] > caps[c]) { fprintf(stderr, "overflow\n"); return 2; }
				}
				free(data); free(offs); free(src);
			}
			xe_pcap_close(p);
		}
	}
	printf("datagrams %lu\n", total);
	return 0;
}
