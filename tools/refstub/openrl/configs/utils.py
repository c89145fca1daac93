class ProcessYamlAction:
    pass
