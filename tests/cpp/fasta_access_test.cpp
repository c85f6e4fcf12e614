// Test driver for panvc3_amd/host/msa_index_file.hpp read_fasta_contigs and bgzf.hpp byte_writer::close
// (tests/test_host_io_cpu.py).
//   fasta_access_test contigs FASTA NAME [NAME ...]   ->  ">NAME\nSEQUENCE\n" per contig found through FASTA.fai (+ .gzi)
//   fasta_access_test full-device                      ->  writes BGZF to /dev/full: close() must throw, exit code 3
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../panvc3_amd/host/msa_index_file.hpp"

int main(int argc, char **argv)
{
	try
	{
		if (argc >= 3 && 0 == std::strcmp(argv[1], "contigs"))
		{
			std::vector <std::string> names(argv + 3, argv + argc);
			std::vector <std::pair <std::string, std::vector <char>>> out;
			if (!pvc3host::read_fasta_contigs(argv[2], names, 2, out)) { std::fprintf(stderr, "no index files\n"); return 2; }
			for (auto const &kv : out)
			{
				std::printf(">%s\n", kv.first.c_str());
				std::fwrite(kv.second.data(), 1, kv.second.size(), stdout);
				std::printf("\n");
			}
			return 0;
		}
		if (argc >= 2 && 0 == std::strcmp(argv[1], "full-device"))
		{
			pvc3host::byte_writer w;
			w.open("/dev/full", true);
			std::string const data(1000, 'x');
			w.write(data);
			try { w.close(); }
			catch (pvc3host::io_error const &e) { std::fprintf(stderr, "%s\n", e.what()); return 3; }
			return 0;
		}
	}
	catch (std::exception const &e)
	{
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 64;
}
