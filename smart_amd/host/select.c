/* select — the registry editor of the harness: which algorithms `smart` runs when no -a list is given.
 *
 * Restates the behaviour of the reference's src/select.c:57-194 over the same file, source/algorithms.h, one
 * "#<0|1> #<name> " line per algorithm (written by select.c:190-193, read by getAlgo, src/function.h:62-77 — and by
 * smart's read_registry, host/smart.c):
 *   select                      "No parameter given"                                      (select.c:67)
 *   select -h                   the manual                                                (select.c:33-46)
 *   select -show                every registered algorithm, one per line                  (select.c:72-80)
 *   select -which               the selected ones                                         (select.c:81-90)
 *   select NAME [NAME ...]      toggles the selection of NAME (exact spelling)            (select.c:159-173)
 *   select -all | -none         selects / deselects everything                            (select.c:175-184)
 *   select -add NAME            registers source/bin/NAME, deselected, after `./test NAME -nv` passed; refuses a
 *                               name that is registered already or a file that is not there (select.c:91-124)
 * Several parameters are processed left to right; -show and -which end the run without writing; an unknown parameter
 * ends it with an error, also without writing (select.c:185).  Otherwise the file is rewritten sorted by name
 * (select.c:186-193).  Plain C: no GPU, no library — the engine enters only through ./test, as in the reference.
 * Not restated: -group (commented out of the reference's manual, a copy of -add in its code, select.c:125-158). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_ALGOS 500 /* NumAlgo, select.c:23 */
#define NAME_LEN 32
#define REGISTRY "source/algorithms.h"

static char names[MAX_ALGOS][NAME_LEN];
static int selected[MAX_ALGOS];
static int count;

static int load(void)
{
    FILE *fp = fopen(REGISTRY, "r");
    if (!fp) return -1;
    int c;
    count = 0;
    while ((c = getc(fp)) != EOF && count < MAX_ALGOS) {
        if (c != '#') continue;
        const int flag = getc(fp);
        if (flag != '0' && flag != '1') continue;
        if (getc(fp) != ' ' || getc(fp) != '#') continue;
        int j = 0;
        while ((c = getc(fp)) != EOF && c != ' ' && c != '\n' && j < NAME_LEN - 1) names[count][j++] = (char)c;
        names[count][j] = 0;
        if (j) selected[count++] = flag - '0';
    }
    fclose(fp);
    return count;
}

static int find(const char *name)
{
    for (int i = 0; i < count; ++i)
        if (!strcmp(names[i], name)) return i;
    return -1;
}

static int find_nocase(const char *name) /* search_ALGO, select.c:48-53 */
{
    for (int i = 0; i < count; ++i) {
        const char *a = names[i], *b = name;
        while (*a && *b && ((*a | 32) == (*b | 32) || *a == *b)) { ++a; ++b; }
        if (!*a && !*b) return i;
    }
    return -1;
}

static int by_name(const void *a, const void *b) { return strcmp(names[*(const int *)a], names[*(const int *)b]); }

static int save(void)
{
    int order[MAX_ALGOS];
    for (int i = 0; i < count; ++i) order[i] = i;
    qsort(order, (size_t)count, sizeof order[0], by_name);
    FILE *fp = fopen(REGISTRY, "w");
    if (!fp) { printf("\n\tSMART error message\n\tcannot write %s\n\n", REGISTRY); return 1; }
    for (int i = 0; i < count; ++i) fprintf(fp, "#%d #%s \n", selected[order[i]], names[order[i]]);
    fclose(fp);
    return 0;
}

static void manual(void)
{
    printf("\n\tSMART UTILITY FOR SELECTING STRING MATCHING ALGORITHMS\n\n");
    printf("\t-show           shows the list of all algorithms\n");
    printf("\t-which          shows the list of all selected algorithms\n");
    printf("\tALGO            selects/deselects the algorithm ALGO (ex. select bf)\n");
    printf("\t-all            selects all algorithms\n");
    printf("\t-none           deselects all algorithms\n");
    printf("\t-add ALGO       add the new alorithm ALGO to the set\n");
    printf("\t                the executable file of the new algorithm must be in /source/bin\n");
    printf("\t-h              gives this help list\n\n\n");
}

int main(int argc, char **argv)
{
    if (argc == 1) { printf("\n\tNo parameter given. Use -h for help.\n\n"); return 0; }
    if (!strcmp(argv[1], "-h")) { manual(); return 0; }
    if (load() < 0) { printf("\n\tSMART error message\n\tcannot read %s\n\n", REGISTRY); return 1; }
    for (int par = 1; par < argc;) {
        const char *arg = argv[par++];
        if (!strcmp(arg, "-show")) {
            printf("The list of all string matching algorithms\n");
            for (int i = 0; i < count; ++i) printf("%s\n", names[i]);
            return 0;
        }
        if (!strcmp(arg, "-which")) {
            printf("\n\tThe list of selected algorithms:\n");
            for (int i = 0; i < count; ++i)
                if (selected[i]) printf("\t-%s\n", names[i]);
            printf("\n");
            return 0;
        }
        if (!strcmp(arg, "-all") || !strcmp(arg, "-none")) {
            for (int i = 0; i < count; ++i) selected[i] = arg[1] == 'a';
            continue;
        }
        if (!strcmp(arg, "-add")) {
            if (par >= argc) { printf("\n\n\tError in input parameters. Use -h for help.\n\n"); return 0; }
            const char *name = argv[par++];
            char path[64 + NAME_LEN], command[64 + NAME_LEN];
            if (strlen(name) >= NAME_LEN || strchr(name, '/') || strchr(name, ' ') || strchr(name, '\'')) {
                printf("\n\n\tSMART error message\n\tError in input parameters....%s is no algorithm name\n\n", name);
                continue;
            }
            snprintf(path, sizeof path, "source/bin/%s", name);
            FILE *fp = fopen(path, "r");
            if (!fp) { printf("\n\n\tSMART error message\n\tError in input parameters....program %s does not exist\n\n", path); continue; }
            fclose(fp);
            if (find_nocase(name) >= 0) {
                printf("\n\n\tSMART error message\n\tError in input parameters....algorithm %s already in the set\n\n", name);
                continue;
            }
            if (count == MAX_ALGOS) { printf("\n\n\tSMART error message\n\tthe set is full (%d algorithms)\n\n", MAX_ALGOS); continue; }
            printf("\n\n\tAdding the algorithm %s to SMART\n\tTesting the algorithm for correctness....", name);
            fflush(stdout);
            snprintf(command, sizeof command, "./test '%s' -nv", name);
            if (system(command)) {
                printf("failed!\n\tThe system is unable to add the algorithm %s to SMART.\n\tPlease, check for algorithm's correctness.\n\n", name);
                continue;
            }
            printf("ok\n");
            strcpy(names[count], name);
            selected[count++] = 0;
            printf("\tAlgorithm %s added succesfully.\n\n", name);
            continue;
        }
        const int i = find(arg);
        if (i < 0) { printf("\tError in input parameters....no parameter %s\n\n", arg); return 0; }
        selected[i] = !selected[i];
        printf("\tThe %s algorithm has been %s\n", names[i], selected[i] ? "selected" : "deselected");
    }
    return save();
}
