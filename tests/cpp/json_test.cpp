// json_test.cpp -- drives radio-sim_amd/host/json.hpp from stdin for tests/test_host_server.py (CPU tier).
//   d <hex bits>    -> Java Double.toString text of that double, then the text a JSON message carries
//   p <json>        -> parse; print the minimal text again, or "error"
//   l <json number> -> asLong of it, or "error"
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "../../radio-sim_amd/host/json.hpp"

int main()
{
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.size() < 2) continue;
        const std::string arg = line.substr(2);
        try {
            if (line[0] == 'd') {
                const unsigned long long bits = std::stoull(arg, nullptr, 16);
                double d;
                std::memcpy(&d, &bits, 8);
                std::string wire;
                try {
                    wire = rsim::json_double_text(d);
                } catch (const rsim::JsonError &) {
                    wire = "refused";
                }
                if (wire != "refused") { // the direct writer must give the same text
                    std::string direct;
                    rsim::append_double(direct, d);
                    if (direct != wire) wire = "MISMATCH:" + direct;
                }
                std::printf("%s %s\n", rsim::java_double_to_string(d).c_str(), wire.c_str());
            } else if (line[0] == 'p') {
                std::printf("%s\n", rsim::Json::parse(arg).toString().c_str());
            } else if (line[0] == 'l') {
                std::printf("%lld\n", (long long)rsim::Json::parse(arg).asLong());
            }
        } catch (const std::exception &) {
            std::printf("error\n");
        }
    }
    return 0;
}
